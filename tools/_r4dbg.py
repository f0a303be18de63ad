import sys, os, torch
sys.path.insert(0, '.')
from locotouch_amd.rl import PPO, ActorCritic, tuned_gemms
from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG
from tests.test_hip_ppo_graph import _fill
mode = sys.argv[1]
if mode == "off": os.environ["LT_TUNED_GEMMS"] = "0"
n, T = 1024, 24
cfg = dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=4)
if mode == "kw": cfg["tuned_gemms"] = False
algs = []
for direct in (True, False):
    torch.manual_seed(0)
    alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", direct_update=direct, **cfg)
    alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
    algs.append(alg)
print(mode, "tunable enabled:", torch.cuda.tunable.is_enabled(), "tuning:", torch.cuda.tunable.tuning_is_enabled())
if mode == "dis":
    tuned_gemms.disable(); print("after disable:", torch.cuda.tunable.is_enabled())
a, b = algs
for it in range(2):
    for alg in (a, b):
        _fill(alg, 200 + it, n, T); torch.manual_seed(11 + it); alg.update()
    d = max(float((pa - pb).abs().max()) for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()))
    g = max(float((pa.grad - pb.grad).abs().max()) for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()))
    print(mode, it, "max param diff", d, "max grad diff", g, a.learning_rate, b.learning_rate)
