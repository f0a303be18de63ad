"""Diagnostic: per-phase cycle shares of lt_step_kernel from in-kernel s_memtime stamps (LT_STAMPS build only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.env import LocoTouchVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = LocoTouchVecEnv("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=n, device="cuda:0")
act = 0.5 * torch.randn(n, 12, device="cuda:0")
for _ in range(50):
    env.step(act)
torch.cuda.synchronize()
st = env.field("LT_F_LAST_EPISODE_SUMS")[::16, :, 0].cpu()  # lane 0 of every wave
names = ["load hot state", "physics (4 substeps)", "late loads + terms + rewards", "reset/command/pushes", "obs frame -> LDS", "history rows", "store state"]
tot = st.sum(1).mean()
for i, nm in enumerate(names):
    print(f"{nm:32s} mean {st[:, i].mean():10.0f} ticks  ({100 * st[:, i].mean() / tot:5.1f} %)   max {st[:, i].max():10.0f}")
bw = env.field("LT_F_REWARD_TERMS")[::16, :2, 0].cpu()
print(f"wave 0 waiting in physics barriers: A {bw[:, 0].mean():.0f} ticks, B {bw[:, 1].mean():.0f} ticks (4 substeps)")
print("total ticks", float(tot), "(s_memtime ticks at 100 MHz => us:", float(tot) / 100.0, ")")
