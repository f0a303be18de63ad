"""Probe: K independent half-populations (env + rollout chain each, own stream, own hipGraph) replayed concurrently - does the chip
overlap one chain's policy launch with another's env step?  Aggregate env-steps/s against the single 4096-env chain."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.env import LocoTouchVecEnv
from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
from bench import POLICY_CFG, PPO_CFG, ROLLOUT, TASKS

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for K in (1, 2, 4):
    n = total // K
    chains = []
    for k in range(K):
        env = LocoTouchVecEnv(TASKS["teacher"], num_envs=n, device="cuda:0", seed=42, env_index_offset=k * n)
        torch.manual_seed(1234)
        alg = PPO(ActorCritic(env.num_obs, env.num_obs, 12, **POLICY_CFG), device="cuda:0", **PPO_CFG)
        alg.init_storage(n, ROLLOUT, [env.num_obs], [env.num_obs], [12])
        fr = FusedRollout(env, alg)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fr.rollout(ROLLOUT)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fr.rollout(ROLLOUT)
        torch.cuda.synchronize()
        chains.append((env, alg, fr, s, g))
    for _ in range(3):
        for c in chains:
            with torch.cuda.stream(c[3]):
                c[4].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        for c in chains:
            with torch.cuda.stream(c[3]):
                c[4].replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"K={K} chains x {n} envs: {total * ROLLOUT * reps / el / 1e6:.1f} M env-steps/s ({1e6 * el / (ROLLOUT * reps):.1f} us per step of the whole population)")
    del chains
