#!/usr/bin/env python3
"""PPO-update GEMM shapes (minibatch 24576 rows, fp32): how fast are the forms torch can issue them in?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

def t(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

M = 24576
print("backend prefer hipblaslt:", os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"))
for (K, N) in [(348, 512), (512, 256), (256, 128), (128, 12), (128, 1)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    Wt = W.t().contiguous(); dy = torch.randn(M, N, device="cuda")
    gf = 2 * M * K * N / 1e9
    a = t(lambda: F.linear(x, W, b)); c = t(lambda: torch.addmm(b, x, Wt)); d = t(lambda: x @ Wt)
    e = t(lambda: dy @ W); f = t(lambda: dy.t() @ x); g = t(lambda: torch.mm(x.t(), dy))
    print(f"K={K:4d} N={N:4d} {gf:6.2f} GF | fwd F.linear {a:7.1f} us ({gf/a*1e3:6.1f} TF/s) | addmm(x,Wt) {c:7.1f} ({gf/c*1e3:6.1f}) | x@Wt {d:7.1f} | "
          f"dX=dy@W {e:7.1f} ({gf/e*1e3:6.1f}) | dW=dy^T@x {f:7.1f} ({gf/f*1e3:6.1f}) | x^T@dy {g:7.1f}")
