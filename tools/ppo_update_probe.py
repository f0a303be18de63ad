#!/usr/bin/env python3
"""Wall time of PPO.update() at the bench shape (4096 envs x 24 steps, 4 minibatches x 5 epochs) with the fused pieces on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.rl import PPO, ActorCritic
from tests.rl_synth import POLICY_CFG, PPO_CFG

dev, n, T, D, A = "cuda:0", 4096, 24, 348, 12


def run(**flags):
    torch.manual_seed(0)
    alg = PPO(ActorCritic(D, D, A, **POLICY_CFG), device=dev, **dict(PPO_CFG, **flags))
    alg.init_storage(n, T, [D], [D], [A])
    st, ac = alg.storage, alg.actor_critic
    g = torch.Generator(device=dev).manual_seed(1)
    times = []
    for it in range(8):
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
        out = alg.update()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    return min(times[2:]) * 1e3, sum(times[2:]) / len(times[2:]) * 1e3, out[:3], alg.learning_rate


for name, flags in (("all torch ops", dict(fused_loss=False, fused_adam=False, packed_forward=False)), ("fused loss", dict(fused_adam=False, packed_forward=False)),
                    ("fused loss + flat Adam", dict(packed_forward=False)), ("+ packed forward (default)", {})):
    if name == "all torch ops":
        from locotouch_amd.rl import linear
        keep = linear.linear_elu_ok
        linear.linear_elu_ok = lambda *a: False
    best, mean, out, lr = run(**flags)
    if name == "all torch ops":
        linear.linear_elu_ok = keep
    print(f"{name:36s} update {best:7.2f} ms best / {mean:7.2f} ms mean  ({best / 20 * 1e3:6.0f} us per minibatch step)  losses {out} lr {lr:.2e}", flush=True)
