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
pr = env.field("LT_F_REWARD_TERMS")[::16, 4:7, 0].cpu()
print(f"prologue: entry -> barrier B0 reached {pr[:, 0].mean():.0f} (max {pr[:, 0].max():.0f}); inside B0 (loads landed + all four waves arrived) {pr[:, 1].mean():.0f} (max {pr[:, 1].max():.0f}); B0 -> wave 0's own path {pr[:, 2].mean():.0f} (max {pr[:, 2].max():.0f})")
ph = env.field("LT_F_REWARD_TERMS")[::16, :7, 1].cpu()  # lane 1 of every wave: phase sums over the step's substeps
if n <= 4096:
    names2 = ["publish + barrier A", "kinematics, foot, guarded spheres, backward pass", "barrier B + fetch + merge", "3x3 Cholesky, Y, Schur update", "quad sums (54 DPP adds)", "6x6 solve", "back-substitution, link accelerations, contact forces"]
    print("substep phases of wave 0 (sum over the step's substeps):")
    for i, nm in enumerate(names2):
        print(f"   {nm:56s} {ph[:, i].mean():8.0f}")
print("total ticks", float(tot), "(s_memtime ticks at 100 MHz => us:", float(tot) / 100.0, ")")

if n <= 4096:  # helper form: when does each wave of a tile finish, relative to its own entry?  (LT_F_REWARD_TERMS quad array 3 carries the stamps)
    rt = env.field("LT_F_REWARD_TERMS").cpu()  # [n][7][4]
    ends = rt[::16, 3, :]  # (wave 0 end, wave 1 end, wave 2 end, wave 3 end)
    print(f"cycles since the wave's entry: wave 0 ends at {ends[:, 0].mean():8.0f} (max {ends[:, 0].max():8.0f}); waves 1 / 2 (history rows) at "
          f"{ends[:, 1].mean():8.0f} / {ends[:, 2].mean():8.0f} (max {ends[:, 1:3].max():8.0f}); wave 3 (noise RNG, curriculum partials) at {ends[:, 3].mean():8.0f} (max {ends[:, 3].max():8.0f})")
