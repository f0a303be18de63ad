#!/usr/bin/env python3
"""Where a BC training batch of the student goes (forward + backward per component) at the distillation's batch shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.distill import Student, distillation_cfg

L, B = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 101
cfg = distillation_cfg("Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1")
cfg.device, cfg.log_dir = "cuda:0", "/tmp"
W = torch.randn(348, 12, device="cuda") * 0.05
st = Student(cfg, 270, 442, 12, teacher_policy_inference=lambda o: o @ W, verbose=False).train()

def t(fn, k=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

masks = torch.ones(L, B, dtype=torch.bool, device="cuda")
batch = dict(proprioceptions=torch.randn(L, B, 270, device="cuda"), teacher_encoder_obses=torch.randn(L, B, 78, device="cuda"),
             tactile_signals=(torch.rand(L, B, 442, device="cuda") < 0.1).float(), masks=masks)
img = batch["tactile_signals"].reshape(L * B, 2, 17, 13)
def fb(mod, inp):
    def f():
        mod.zero_grad()
        x = inp.clone().requires_grad_(True)
        mod(x).sum().backward()
    return f
print(f"L={L} B={B} ({L * B} padded steps)")
print("pre_encoder (CNN head) fwd+bwd ms", t(fb(st.pre_encoder, img)))
emb = torch.randn(L, B, 64, device="cuda")
print("student_encoder (GRU + MLP) fwd+bwd ms", t(fb(st.student_encoder, emb)))
print("student_backbone fwd+bwd ms", t(fb(st.student_backbone, torch.randn(L, B, 334, device="cuda"))))
def full():
    st._optimizer.zero_grad(set_to_none=True)
    loss, _, _ = st.batch_loss(batch)
    loss.backward()
    st._optimizer.step()
print("full training step ms", t(full))
for name, m in (("conv stack only", st.pre_encoder.conv), ("head only", st.pre_encoder.head)):
    inp = img if name.startswith("conv") else torch.randn(L * B, 192, device="cuda")
    print(name, "fwd+bwd ms", t(fb(m, inp)))
