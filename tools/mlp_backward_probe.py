#!/usr/bin/env python3
"""Time of lt_mlp_backward_pair alone (both stacks of the LocoTouch ActorCritic, m rows) - and of probe builds without the gate loads /
the dz stores (tools/build_variant.py <name> -DLT_GATE_NO_LOAD ...; LOCOTOUCH_AMD_LIB selects the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.rl import mlp as M
from tests.test_hip_mlp_backward import _nets

m = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
actor, critic = _nets()
pair = M.PackedPair(actor, critic)
x0, x1 = torch.randn(m, 348, device="cuda"), torch.randn(m, 348, device="cuda")
dy0, dy1 = torch.randn(m, 12, device="cuda") * 1e-4, torch.randn(m, 1, device="cuda") * 1e-4
grads = {p: torch.zeros_like(p) for net in (actor, critic) for p in net.parameters()}
_, acts = pair.forward_raw(x0, x1)
import ctypes
lib, vp = M._abi.load(), ctypes.c_void_p
nets = (pair.a, pair.b)
for n in nets: n.pack_backward()
nblk = int(lib.lt_mlp_backward_blocks(ctypes.byref(nets[0].desc), ctypes.byref(nets[1].desc), m))
dzs = [[torch.empty_like(a) for a in acts[k]] for k in range(2)]
ams = [torch.empty(3, nblk, device="cuda") for k in range(2)]
arr = ctypes.c_void_p * 3
arrs = [(arr(*[a.data_ptr() for a in acts[k]]), arr(*[t.data_ptr() for t in dzs[k]]), arr(*[ams[k][l].data_ptr() for l in range(3)])) for k in range(2)]
sat = torch.zeros(1, device="cuda")
def run():
    M._abi.check(lib.lt_mlp_backward_pair(ctypes.byref(nets[0].desc), vp(nets[0].bpacked.data_ptr()), vp(dy0.data_ptr()), *arrs[0],
                                          ctypes.byref(nets[1].desc), vp(nets[1].bpacked.data_ptr()), vp(dy1.data_ptr()), *arrs[1],
                                          m, int(pair.acts_split), vp(None), vp(None), 0, vp(None), vp(sat.data_ptr()), M.PackedMLP._stream()), "bwd")
def fwd():
    pair.forward_raw(x0, x1)
for name, fn in (("backward chain", run), ("forward pair", fwd)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(f"{os.environ.get('LOCOTOUCH_AMD_LIB', 'product')[-24:]:24s} {name}: {(time.perf_counter() - t0) / 50 * 1e6:7.1f} us  (m = {m})", flush=True)
