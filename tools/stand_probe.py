"""Diagnostic: zero-action stand - which termination fires, when, and what the base / cylinder do until then."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from locotouch_amd.env import LocoTouchVecEnv, TERMINATION_NAMES
for task in ("Isaac-Locomotion-LocoTouch-v1", "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"):
    n = 256
    e0 = LocoTouchVecEnv(task, num_envs=n, device="cuda:0", seed=3, enable_corruption=0)
    cfg = e0.cfg
    for r in (cfg.push_robot_interval, cfg.push_obj_interval):
        r[0] = r[1] = 1e9
    env = LocoTouchVecEnv(task, device="cuda:0", cfg=cfg)
    zero = torch.zeros(n, 12, device="cuda:0")
    first = torch.full((n,), -1, dtype=torch.long, device="cuda:0")
    bits_at = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    bits = env.view(env._lib and __import__("locotouch_amd")._abi.CONSTS["LT_F_TERM_BITS"])
    for t in range(999):
        env.step(zero)
        tm = env.terminated_buf.bool() & (first < 0)
        first[tm] = t
        bits_at[tm] = bits[tm]
        if t in (0, 5, 20, 50, 100, 200, 400, 800):
            rp = env.field("LT_F_ROOT_POS")[0, 0].cpu().numpy(); rq = env.field("LT_F_ROOT_QUAT")[0, 0].cpu().numpy()
            op = env.field("LT_F_OBJ_POS")[0, 0].cpu().numpy(); q = env.field("LT_F_JOINT_POS")[0].cpu().numpy().reshape(-1)
            fh = env.field("LT_F_FORCE_HIST")[0].cpu().numpy().reshape(-1)[12:16]
            print(f"{task[6:20]} t={t:4d} root z {rp[2]:.4f} quat {np.round(rq,4)} obj-rel {np.round(op[:3]-rp[:3],4)} foot|F| {np.round(fh,1)} q {np.round(q,3)}")
    f = first.cpu().numpy(); b = bits_at.cpu().numpy()
    print(task, "terminated", int((f >= 0).sum()), "/", n, "first-termination step quantiles", np.quantile(f[f >= 0], [0, .25, .5, .75, 1]) if (f >= 0).any() else None)
    for i, nm in enumerate(TERMINATION_NAMES):
        print("   ", nm, int(((b >> i) & 1).sum()))
