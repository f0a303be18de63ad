"""Diagnostic: graph-replayed vs eager PPO update over a few iterations (epochs, minibatches as argv)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from locotouch_amd.rl import PPO, ActorCritic
from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG
n, T = 512, 24
E, MB = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(PPO_CFG, num_learning_epochs=E, num_mini_batches=MB)
if len(sys.argv) > 3: cfg["schedule"] = sys.argv[3]
def make(graph):
    torch.manual_seed(0)
    alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", graph_update=graph, **cfg)
    alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT]); return alg
def fill(alg, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed); ac = alg.actor_critic
    with torch.no_grad():
        for t in range(T):
            o = torch.randn(n, N_OBS, device="cuda:0", generator=g); c = o + 0.01 * torch.randn(n, N_OBS, device="cuda:0", generator=g)
            ac.update_distribution(o)
            act = ac.action_mean + ac.action_std * torch.randn(n, N_ACT, device="cuda:0", generator=g)
            alg._t = dict(actions=act, values=ac.evaluate(c), log_prob=ac.get_actions_log_prob(act), mu=ac.action_mean.clone(), sigma=ac.action_std.clone(), obs=o, critic_obs=c)
            alg.process_env_step(torch.randn(n, device="cuda:0", generator=g), (torch.rand(n, device="cuda:0", generator=g) < 0.05).long(), {})
        alg.compute_returns(torch.randn(n, N_OBS, device="cuda:0", generator=g))
from locotouch_amd.rl.linear import Linear
if os.environ.get("SPLITK_MIN"): Linear.split_k_min_rows = int(os.environ["SPLITK_MIN"])
a, b = make(True), make(False)
import numpy as np
_orig = b._adapt_learning_rate
def _f32(*args):
    _orig(*args)
    b.learning_rate = float(np.float32(b.learning_rate))
    for g in b.optimizer.param_groups: g["lr"] = b.learning_rate
if os.environ.get("F32LR"): b._adapt_learning_rate = _f32
for it in range(4):
    outs = []
    for alg in (a, b):
        fill(alg, 100 + it)
        if it >= 1:
            d = max(float((x - y).abs().max()) for x, y in zip([a.storage.observations, a.storage.advantages, a.storage.returns, a.storage.values, a.storage.mu], [b.storage.observations, b.storage.advantages, b.storage.returns, b.storage.values, b.storage.mu])) if alg is b else None
            if d is not None: print("it", it, "storage max diff a vs b", d)
        torch.manual_seed(7 + it)
        outs.append(alg.update())
    pd = max(float((x - y).abs().max()) for x, y in zip(a.actor_critic.parameters(), b.actor_critic.parameters()))
    steps = [float(s["step"]) for s in list(a.optimizer.state.values())[:1]], [float(s["step"]) for s in list(b.optimizer.state.values())[:1]]
    print("it", it, "outs", [tuple(round(v, 5) for v in o[:3]) for o in outs], "lr", a.learning_rate, b.learning_rate, "param max diff", pd, "adam steps", steps)
