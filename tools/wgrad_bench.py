#!/usr/bin/env python3
"""Time of lt_wgrad alone (no slab sum) for the three wide layers of the LocoTouch networks at M rows, x in the split format and
as f32 rows.  LT_WGRAD_TILED=0 selects the one-wave-per-tile kernel, LT_WGRAD_TILED_BLOCKS the tiled kernel's workgroup target."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd import _abi

lib, vp = _abi.load(), ctypes.c_void_p
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
tag = f"tiled={os.environ.get('LT_WGRAD_TILED', 'auto')} blocks={os.environ.get('LT_WGRAD_TILED_BLOCKS', '512')}"
for (n, k) in [(512, 348), (256, 512), (128, 256)]:
    dz = torch.randn(M, n, device="cuda") / M
    x = torch.randn(M, k, device="cuda")
    xs = torch.empty_like(x)
    st = vp(torch.cuda.current_stream().cuda_stream)
    _abi.check(lib.lt_split_rows(vp(x.data_ptr()), vp(xs.data_ptr()), x.numel(), st), "split")
    am = dz.abs().max().reshape(1)
    sp = int(lib.lt_wgrad_splits(M, n, k))
    slabs = torch.empty(sp * n * k + sp * n, device="cuda")
    ref = (dz.double().t() @ x.double())
    # dz in the split format as the backward chain writes it: scaled by a power of two (max -> [1, 2)), then (hi | lo) dwords
    import math
    sc = torch.tensor([2.0 ** (-math.floor(math.log2(float(dz.abs().max()))))], device="cuda")
    dzs = torch.empty_like(dz)
    _abi.check(lib.lt_split_rows(vp((dz * sc).data_ptr()), vp(dzs.data_ptr()), dz.numel(), st), "split dz")
    run2 = lambda: _abi.check(lib.lt_wgrad(vp(dzs.data_ptr()), 1, vp(sc.data_ptr()), vp(xs.data_ptr()), 1, M, n, k, vp(None), 0, vp(slabs.data_ptr()),
                                           vp(slabs[sp * n * k:].data_ptr()), st), "lt_wgrad")
    for _ in range(5): run2()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): run2()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
    got = slabs[:sp * n * k].view(sp, n, k).double().sum(0)
    db = slabs[sp * n * k:].view(sp, n).double().sum(0)
    print(f"[{tag} deep={os.environ.get('LT_WGRAD_DEEP', '0')}] dW {n} x {k} (split dz, split x): {us:6.1f} us, {sp} slices, err {float((got - ref).abs().max() / ref.abs().max()):.1e}, "
          f"db err {float((db - dz.double().sum(0)).abs().max() / dz.double().sum(0).abs().max()):.1e}", flush=True)
    for name, xx, flag in (("split x", xs, 1), ("f32 x", x, 0)):
        run = lambda: _abi.check(lib.lt_wgrad(vp(dz.data_ptr()), 0, vp(None), vp(xx.data_ptr()), flag, M, n, k, vp(am.data_ptr()), 1, vp(slabs.data_ptr()),
                                              vp(slabs[sp * n * k:].data_ptr()), st), "lt_wgrad")
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): run()
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
        got = slabs[:sp * n * k].view(sp, n, k).double().sum(0)
        db = slabs[sp * n * k:].view(sp, n).double().sum(0)
        err = float((got - ref).abs().max() / ref.abs().max())
        dberr = float((db - dz.double().sum(0)).abs().max() / dz.double().sum(0).abs().max())
        print(f"[{tag}] dW {n} x {k} ({name}): {us:6.1f} us, {sp} slices, {2 * M * n * k / us * 1e-6:6.1f} TF f32-equiv, err {err:.1e}, db err {dberr:.1e}", flush=True)
