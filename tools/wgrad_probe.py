#!/usr/bin/env python3
"""Weight-gradient GEMM dW[N,K] = dy[M,N]^T @ x[M,K] with M = 24576: one skinny GEMM vs split-K as a batched GEMM + sum."""
import os, sys, time
import torch

def t(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

M = 24576
for (K, N) in [(348, 512), (512, 256), (256, 128), (128, 12), (128, 1)]:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
    ref = dy.t() @ x
    base = t(lambda: dy.t() @ x)
    line = f"K={K:4d} N={N:4d} plain {base:7.1f} us |"
    for S in (4, 8, 16, 32, 64, 128):
        def f():
            return torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0)
        err = float((f() - ref).abs().max() / ref.abs().max())
        line += f" S={S}: {t(f):6.1f} (err {err:.1e})"
    print(line)
