#!/usr/bin/env python3
"""lt_partial_sums alone on the job set of one PPO optimizer step (slabs + bias partials of six layers, two head blocks), hot and cold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.rl.mlp import SumJobs

def jobs(which):
    J = SumJobs(); keep = []
    for net in range(2):
        for (n, k, sp) in ((512, 348, 10), (256, 512, 16), (128, 256, 64)):
            if "slabs" in which:
                ws = torch.randn(sp * n * k, device="cuda"); out = torch.empty(n, k, device="cuda"); J.add(ws, sp, n * k, n * k, n * k, out); keep += [ws, out]
            if "db" in which:
                ws = torch.randn(sp * n, device="cuda"); out = torch.empty(n, device="cuda"); J.add(ws, sp, n, n, n, out); keep += [ws, out]
        if "head" in which:
            n = 12 if net == 0 else 1
            ws = torch.randn(256 * (n * 128 + 16), device="cuda"); o0 = torch.empty(n, 128, device="cuda"); o1 = torch.empty(n, device="cuda")
            J.add(ws, 256, n * 128 + 16, n * 128 + n, n * 128, o0, o1); keep += [ws, o0, o1]
    return J, keep

for which in (("slabs", "db", "head"), ("slabs",), ("db",), ("head",), ("slabs", "db")):
    J, keep = jobs(which)
    saved = list(J.jobs)
    def run():
        J.jobs = list(saved); J.launch()
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize(); hot = (time.perf_counter() - t0) / 50 * 1e6
    big = torch.empty(300 * 1024 * 1024 // 4, device="cuda")
    ts = []
    for _ in range(10):
        big.normal_()  # flush L2 / Infinity Cache
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print(f"{'+'.join(which):18s} hot {hot:6.1f} us   cold (incl. launch + sync) {min(ts):6.1f} us", flush=True)
