#!/usr/bin/env python3
"""Record library-GEMM algorithm choices for the PPO update's GEMM shapes with PyTorch TunableOp (run on the target GPU):

    python tools/tune_gemms.py [--envs 4096] [--out gpurun_out/tunableop_gfx950.csv]

runs a few PPO iterations of the headline configuration with tuning ON and writes the results file; copy it to
locotouch_amd/rl/tunableop_gfx950.csv (rl/tuned_gemms.py reads it with tuning OFF).  The file's Validator lines bind it to the
PyTorch / ROCm / hipBLASLt / rocBLAS builds and the GPU architecture it was recorded on."""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[4096])
ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "tunableop_gfx950.csv"))
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()
os.environ["LT_TUNED_GEMMS"] = "0"
import torch
t = torch.cuda.tunable
os.makedirs(os.path.dirname(args.out), exist_ok=True)
t.set_filename(args.out, insert_device_ordinal=False)
t.set_max_tuning_duration(30)
t.set_max_tuning_iterations(30)
t.enable(True)
t.tuning_enable(True)
from locotouch_amd.env import LocoTouchVecEnv
from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
import bench as B
for n in args.envs:
    env = LocoTouchVecEnv(B.TASKS["teacher"], num_envs=n, device="cuda:0", seed=42)
    torch.manual_seed(1234)
    alg = PPO(ActorCritic(env.num_obs, env.num_obs, 12, **B.POLICY_CFG), device="cuda:0", **B.PPO_CFG)
    alg.init_storage(n, B.ROLLOUT, [env.num_obs], [env.num_obs], [12])
    fused = FusedRollout(env, alg)
    _, extras = env.get_observations()
    t0 = time.time()
    for it in range(args.iters):
        fused.rollout(B.ROLLOUT)
        with torch.inference_mode():
            alg.compute_returns(extras["observations"]["critic"])
        alg.update()
        torch.cuda.synchronize()
        print(f"envs {n} iteration {it}: {time.time() - t0:.1f} s", flush=True)
ok = t.write_file(args.out)
print("wrote", args.out, ok)
for r in t.get_results():
    print(r)
