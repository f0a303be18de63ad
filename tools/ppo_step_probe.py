#!/usr/bin/env python3
"""Where one PPO minibatch step (24 576 rows) goes on the GPU: gather / forward / loss chain / backward / clip + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.rl import PPO, ActorCritic
from tests.rl_synth import POLICY_CFG, PPO_CFG

dev = "cuda:0"
M, D, A = 24576, 348, 12
torch.manual_seed(0)
alg = PPO(ActorCritic(D, D, A, **POLICY_CFG), device=dev, **PPO_CFG)
ac = alg.actor_critic
flat = dict(obs=torch.randn(4 * M, D, device=dev), cobs=torch.randn(4 * M, D, device=dev), actions=torch.randn(4 * M, A, device=dev),
            values=torch.randn(4 * M, 1, device=dev), adv=torch.randn(4 * M, 1, device=dev), returns=torch.randn(4 * M, 1, device=dev),
            logp=torch.randn(4 * M, 1, device=dev), mu=torch.randn(4 * M, A, device=dev), sigma=torch.ones(4 * M, A, device=dev))
idx = torch.randperm(4 * M, device=dev)[:M]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
acc = [0.0] * 6
K = 30
FUSED = "--fused" in sys.argv
from locotouch_amd.rl.fused_loss import fused_ppo_loss
for it in range(K + 5):
    ev[0].record()
    b = {k: v[idx] for k, v in flat.items()}
    ev[1].record()
    if FUSED:
        mu_, value = ac.actor(b["obs"]), ac.critic(b["cobs"])
        ev[2].record()
        loss, surrogate_loss, value_loss, ent, kl_mean = fused_ppo_loss(mu_, ac.std, value, b["actions"], b["logp"], b["adv"], b["returns"], b["values"],
                                                                        b["mu"], b["sigma"], 0.2, 1.0, 0.01, True)
    else:
        ac.update_distribution(b["obs"]); value = ac.evaluate(b["cobs"])
        ev[2].record()
    if not FUSED:
      log_prob = ac.get_actions_log_prob(b["actions"]); mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
      with torch.inference_mode():
        kl = torch.sum(torch.log(sigma / b["sigma"] + 1.0e-5) + (torch.square(b["sigma"]) + torch.square(b["mu"] - mu)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1)
        kl_mean = torch.mean(kl)
      ratio = torch.exp(log_prob - torch.squeeze(b["logp"])); a = torch.squeeze(b["adv"])
      surrogate_loss = torch.max(-a * ratio, -a * torch.clamp(ratio, 0.8, 1.2)).mean()
      clipped = b["values"] + (value - b["values"]).clamp(-0.2, 0.2)
      value_loss = torch.max((value - b["returns"]).pow(2), (clipped - b["returns"]).pow(2)).mean()
      loss = surrogate_loss + value_loss - 0.01 * entropy.mean()
    ev[3].record()
    alg.optimizer.zero_grad()
    loss.backward()
    ev[4].record()
    torch.nn.utils.clip_grad_norm_(ac.parameters(), 1.0)
    alg.optimizer.step()
    ev[5].record()
    _ = float(kl_mean) + float(value_loss) + float(surrogate_loss)
    ev[6].record()
    torch.cuda.synchronize()
    if it >= 5:
        for i in range(6):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
names = ["gather 9 tensors", "actor + critic forward", "loss chain (log-prob, KL, surrogate, value, entropy)", "backward (incl. loss-chain backward)", "clip + Adam", "host reads"]
for n, a in zip(names, acc):
    print(f"{n:56s} {a / K * 1e3:8.1f} us")
print("total", sum(acc) / K * 1e3, "us")
