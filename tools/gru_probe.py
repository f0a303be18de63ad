#!/usr/bin/env python3
"""GRU over whole trajectories at the distillation's shape: MIOpen (nn.GRU) vs rl/gru.py, forward + backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from locotouch_amd.rl.gru import gru_sequence

def t(fn, k=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

for (L, B) in [(500, 48), (500, 128), (100, 256), (24, 822)]:
    gru = nn.GRU(64, 512).cuda()
    x = torch.randn(L, B, 64, device="cuda", requires_grad=True)
    def run(fn):
        def f():
            gru.zero_grad(); x.grad = None
            out, _ = fn(x)
            out.sum().backward()
        return f
    import locotouch_amd.rl.gru as G
    a = t(run(lambda v: gru(v)))
    G.use_hip_kernels = False
    b = t(run(lambda v: gru_sequence(gru, v)))
    G.use_hip_kernels = True
    c = t(run(lambda v: gru_sequence(gru, v)))
    print(f"L={L:4d} B={B:4d}: nn.GRU (MIOpen) {a:8.2f} ms | torch-op loop {b:8.2f} ms | lt_gru.hip {c:8.2f} ms | x{a / c:5.1f}   ({L * B / c * 1e3 / 1e6:.2f} M steps/s)", flush=True)
