#!/usr/bin/env python3
"""PPO.update() at the bench shape (4096 envs x 24 steps, 5 epochs x 4 minibatches = 20 optimizer steps) for a kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/ppo_prof -o ppo -- python3 tools/ppo_update_profile.py
    python3 tools/ppo_update_profile.py --summarise gpurun_out/ppo_prof/ppo_kernel_stats.csv > profiles/rNN_ppo_step_timeline.txt
The summary divides every kernel's total time by the number of optimizer steps traced."""
import csv, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ITERS, STEPS = 6, 20

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = list(csv.DictReader(open(sys.argv[2])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {ITERS} updates x {STEPS} optimizer steps (+ the synthetic rollout fill between them); us per optimizer step = total / {ITERS * STEPS}")
    print(f"# all kernels: {tot / (ITERS * STEPS) / 1e3:.1f} us of GPU time per optimizer step")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        t, c = float(r["TotalDurationNs"]), int(r["Calls"])
        print(f"{t / (ITERS * STEPS) / 1e3:8.1f} us/step  {c / (ITERS * STEPS):6.2f} calls/step  {t / c / 1e3:8.1f} us avg  {r['Name'][:110]}")
    sys.exit(0)

import torch
from locotouch_amd.rl import PPO, ActorCritic
from tests.rl_synth import POLICY_CFG, PPO_CFG

dev, n, T, D, A = "cuda:0", 4096, 24, 348, 12
torch.manual_seed(0)
alg = PPO(ActorCritic(D, D, A, **POLICY_CFG), device=dev, **PPO_CFG)
alg.init_storage(n, T, [D], [D], [A])
st, ac = alg.storage, alg.actor_critic
g = torch.Generator(device=dev).manual_seed(1)
times = []
for it in range(ITERS):
    with torch.no_grad():
        st.observations.normal_(generator=g); st.privileged_observations.copy_(st.observations)
        flat = st.observations.flatten(0, 1)
        ac.update_distribution(flat)
        act = ac.distribution.sample()
        st.actions.copy_(act.view(T, n, A)); st.mu.copy_(ac.action_mean.view(T, n, A)); st.sigma.copy_(ac.action_std.view(T, n, A))
        st.actions_log_prob.copy_(ac.get_actions_log_prob(act).view(T, n, 1)); st.values.copy_(ac.evaluate(flat).view(T, n, 1))
        st.rewards.normal_(generator=g); st.dones.zero_(); st.step = T
        alg.compute_returns(st.privileged_observations[-1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    alg.update()
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
print("update ms:", " ".join(f"{t:.2f}" for t in times), flush=True)
