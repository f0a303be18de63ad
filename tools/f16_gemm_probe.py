#!/usr/bin/env python3
"""PPO-update backward GEMMs as f16 library GEMMs on pre-split (hi, lo) operand planes with f32 output (torch.mm(..., out_dtype=float32)):
is the route faster than the f32 GEMMs the update runs today?  Shapes: minibatch M = 24576, layers 348(352)-512-256-128.

  dX form:  [64 hi | lo | hi] (M x 3N)  @  [W_hi ; W_hi ; W_lo] (3N x K)              -> M x K     (alpha = 1/64)
  dW form:  [hi | lo]^T (2N x M)        @  [x_hi | x_lo] (M x 2K)                     -> 2N x 2K   (four blocks, combined afterwards)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def t(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

M = 24576
dev = "cuda"
print("torch", torch.__version__)
ok = True
try:
    a = torch.randn(64, 64, device=dev).half(); b = torch.randn(64, 64, device=dev).half()
    c = torch.mm(a, b, out_dtype=torch.float32)
    print("mm out_dtype ok:", c.dtype, float((c - a.float() @ b.float()).abs().max()))
except Exception as e:
    ok = False
    print("mm out_dtype FAILED:", type(e).__name__, str(e)[:300])
if ok:
    for (N, K) in [(256, 512), (128, 256)]:
        A = torch.randn(M, 3 * N, device=dev).half(); B = torch.randn(3 * N, K, device=dev).half()
        Af = torch.randn(M, N, device=dev); Bf = torch.randn(N, K, device=dev)
        gf = 2 * M * N * K / 1e9
        x = t(lambda: torch.mm(A, B, out_dtype=torch.float32)); y = t(lambda: Af @ Bf)
        Bt = B.t().contiguous()
        x2 = t(lambda: torch.mm(A, Bt.t(), out_dtype=torch.float32))
        print(f"dX N={N} K={K}: f16 planes (K'=3N) {x:7.1f} us = {gf/x*1e3:6.1f} TF f32-equiv ({3*gf/x*1e3:6.1f} TF f16) | B^T layout {x2:7.1f} | f32 gemm {y:7.1f} us = {gf/y*1e3:6.1f} TF")
    for (N, K) in [(512, 352), (256, 512), (128, 256)]:
        D = torch.randn(M, 2 * N, device=dev).half(); X = torch.randn(M, 2 * K, device=dev).half()
        Df = torch.randn(M, N, device=dev); Xf = torch.randn(M, K, device=dev)
        gf = 2 * M * N * K / 1e9
        one = t(lambda: torch.mm(D.t(), X, out_dtype=torch.float32))
        res = [f"one mm {one:7.1f}"]
        for S in (4, 8, 16, 32):
            Db = D.view(S, M // S, 2 * N); Xb = X.view(S, M // S, 2 * K)
            try:
                tt = t(lambda: torch.bmm(Db.transpose(1, 2), Xb, out_dtype=torch.float32).sum(0))
                res.append(f"bmm/{S} {tt:7.1f}")
            except Exception as e:
                res.append(f"bmm/{S} FAILED {type(e).__name__}")
        f32 = t(lambda: Df.t() @ Xf)
        Dfb = Df.view(8, M // 8, N); Xfb = Xf.view(8, M // 8, K)
        f32s = t(lambda: torch.bmm(Dfb.transpose(1, 2), Xfb).sum(0))
        print(f"dW N={N} K={K} ({gf:5.1f} GF f32-equiv): " + " | ".join(res) + f" | f32 one {f32:7.1f} | f32 bmm/8 {f32s:7.1f} us")
    # 3-plane dW alternative: [64hi|lo|hi] is what dX wants; its [lo|hi] tail is the dW operand (a view with a row stride of 3N)
    N, K = 256, 512
    D3 = torch.randn(M, 3 * N, device=dev).half(); X = torch.randn(M, 2 * K, device=dev).half()
    v = D3[:, N:]
    print("dW on a strided view:", t(lambda: torch.mm(v.t(), X, out_dtype=torch.float32)))
    Db = v.reshape(8, M // 8, 2 * N) if v.is_contiguous() else None
    # elementwise costs around the GEMMs
    g = torch.randn(M, 512, device=dev)
    print("f32 add of two M x 512:", t(lambda: g + g), "us;  half cast M x 512:", t(lambda: g.half()))
