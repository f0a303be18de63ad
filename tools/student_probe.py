#!/usr/bin/env python3
"""Where the student's single-step inference time goes (405 envs): CNN head, GRU step (MIOpen nn.GRU vs fused gru_cell), MLPs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.distill import Student, distillation_cfg

cfg = distillation_cfg("Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1")
cfg.device, cfg.log_dir = "cuda:0", "/tmp"
st = Student(cfg, 270, 442, 12, verbose=False).eval()
n = 405
prop, tac = torch.randn(n, 270, device="cuda"), (torch.rand(n, 442, device="cuda") < 0.1).float()

def t(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

with torch.no_grad():
    img = tac.reshape(n, 2, 17, 13)
    print("pre_encoder (CNN head) us", t(lambda: st.pre_encoder(img)))
    emb = st.pre_encoder(img)
    gru = st.student_encoder.memory.rnn
    h = torch.zeros(1, n, 512, device="cuda")
    print("nn.GRU step us", t(lambda: gru(emb.unsqueeze(0), h)))
    print("gru_cell us", t(lambda: torch.gru_cell(emb, h[0], gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)))
    a, _ = gru(emb.unsqueeze(0), h); b = torch.gru_cell(emb, h[0], gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    print("max diff", float((a[0] - b).abs().max()))
    print("encoder mlp us", t(lambda: st.student_encoder.mlp(b)))
    print("backbone us", t(lambda: st.student_backbone(torch.cat((prop, st.student_encoder.mlp(b)), -1))))
    print("full forward us", t(lambda: st(prop, tac)))
