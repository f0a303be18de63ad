"""Helper of test_hip_forms.py: step a seeded env K times with fixed actions and dump what a caller sees (own process: the step
kernel's form is chosen once per process, LT_STEP_HELPERS_MAX_WG).  python -m tests.form_dump <task id> <n> <steps> <out.npz>"""
import sys

import numpy as np
import torch

from locotouch_amd.env import LocoTouchVecEnv

task, n, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
env = LocoTouchVecEnv(task, num_envs=n, device="cuda:0", seed=23)
env.reset()
g = torch.Generator(device="cpu").manual_seed(5)
rec = {}
for t in range(steps):
    act = (0.6 * torch.randn(n, 12, generator=g)).to("cuda:0")
    obs, rew, dones, extras = env.step(act)
    rec[f"obs{t}"] = obs.cpu().numpy().copy()
    rec[f"critic{t}"] = extras["observations"]["critic"].cpu().numpy().copy()
    rec[f"rew{t}"] = rew.cpu().numpy().copy()
    rec[f"done{t}"] = dones.cpu().numpy().copy()
torch.cuda.synchronize()
for f in ("LT_F_ROOT_POS", "LT_F_JOINT_POS", "LT_F_JOINT_VEL", "LT_F_CMD", "LT_F_EPISODE_SUMS"):
    rec[f] = env.field(f).cpu().numpy().copy()
np.savez(out, **rec)
