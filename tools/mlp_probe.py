"""Diagnostic: time of the fused MLP launch against the torch module it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.rl.mlp import PackedMLP
from locotouch_amd.rl.modules import build_mlp

import ast
dims = ast.literal_eval(os.environ.get("DIMS", "(348, 512, 256, 128, 12)"))
for m in [int(a) for a in sys.argv[1:]] or [4096]:
    seq = build_mlp(dims[0], list(dims[1:-1]), dims[-1], "elu").to("cuda:0")
    x = torch.randn(m, dims[0], device="cuda:0")
    net = PackedMLP(seq)
    out = torch.empty(m, dims[-1], device="cuda:0")
    def timeit(f, n=200):
        for _ in range(20): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    with torch.inference_mode():
        t_f = timeit(lambda: net(x, out))
        t_t = timeit(lambda: seq(x))
    flops = 2 * m * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    print(f"m={m}: fused {t_f:.1f} us ({flops / t_f / 1e6:.1f} TFLOP/s)   torch {t_t:.1f} us ({flops / t_t / 1e6:.1f} TFLOP/s)")

    lib = net.lib
    if hasattr(lib, "lt_debug_mlp_stamps"):
        import ctypes, numpy as np
        net(x, out); torch.cuda.synchronize()
        NWV = int(os.environ.get("NWV", "4"))
        buf = (ctypes.c_uint64 * (1024 * 8 * NWV))()
        lib.lt_debug_mlp_stamps(buf)
        nb = min(1024, (m + 15) // 16 // int(os.environ.get("LT_MLP_ROW_TILES", "1")))
        st = np.array(buf, dtype=np.uint64).reshape(1024, NWV, 8)[:nb].astype(np.int64)
        L = len(dims) - 1
        t0 = st[:, :, 0].min()
        print("stamps (10 ns ticks -> ns), mean over blocks, per wave 0..3; times since kernel start")
        names = ["start", "input staged"] + [f"layer {l} done" for l in range(L)]
        for i, nm in enumerate(names):
            print(f"   {nm:16s}", " ".join(f"{(st[:, w, i] - t0).mean() * 10:8.0f}" for w in range(NWV)))
        print(f"   {'Lx barrier passed':16s}", " ".join(f"{(st[:, w, 7] - t0).mean() * 10:8.0f}" for w in range(NWV)))
        print(f"   {'Lx loop done':16s}", " ".join(f"{(st[:, w, 6] - t0).mean() * 10:8.0f}" for w in range(NWV)))
        print("   last block end - first start:", (st[:, :, L + 1].max() - t0) * 10, "ns")
