#!/usr/bin/env python3
"""Does a NEW batch shape cost extra (MIOpen find / workspace / allocator)?  First-call vs repeat-call times per component."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.distill import Student, distillation_cfg
cfg = distillation_cfg("Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1")
cfg.device, cfg.log_dir = "cuda:0", "/tmp"
W = torch.randn(348, 12, device="cuda") * 0.05
st = Student(cfg, 270, 442, 12, teacher_policy_inference=lambda o: o @ W, verbose=False).train()
def once(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def fb(mod, inp):
    def f():
        mod.zero_grad(); x = inp.clone().requires_grad_(True); mod(x).sum().backward()
    return f
for (L, B) in [(500, 101), (487, 101), (500, 101), (455, 97)]:
    img = (torch.rand(L * B, 2, 17, 13, device="cuda") < 0.1).float()
    emb = torch.randn(L, B, 64, device="cuda")
    f1, f2 = fb(st.pre_encoder, img), fb(st.student_encoder, emb)
    print(f"L={L} B={B}: CNN first {once(f1):8.1f} ms, again {once(f1):7.1f} | GRU+MLP first {once(f2):8.1f} ms, again {once(f2):7.1f}", flush=True)
