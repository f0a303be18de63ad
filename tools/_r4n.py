import torch, os, sys
sys.path.insert(0, os.getcwd())
from locotouch_amd.rl import PPO, ActorCritic, tuned_gemms
from locotouch_amd.rl import mlp as M
from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG
from tests.test_hip_ppo_graph import _fill
tuned_gemms.disable()
n, T = 1024, 24
cfg = dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=4, tuned_gemms=False)
orig = M.PackedPair.backward_raw
def synced(self, *a):
    torch.cuda.synchronize(); orig(self, *a); torch.cuda.synchronize()
def run(direct, flag, sync=False):
    M.PackedPair.backward_raw = synced if sync else orig
    M.USE_SPLIT_F16_WGRAD = flag
    torch.manual_seed(0)
    alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", direct_update=direct, **cfg)
    alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
    _fill(alg, 200, n, T)
    torch.manual_seed(11)
    alg.update()
    return {k: (p.detach().clone(), p.grad.clone()) for k, p in alg.actor_critic.named_parameters()}
ref = run(False, False)
for name, args in (("direct+wgrad", (True, True)), ("direct+wgrad again", (True, True)), ("direct+wgrad synced", (True, True, True)), ("direct+lib", (True, False)), ("direct+lib synced", (True, False, True))):
    r = run(*args)
    wg = max((float((r[k][1] - ref[k][1]).abs().max() / ref[k][1].abs().max()), k) for k in ref)
    wp = max((float((r[k][0] - ref[k][0]).abs().max()), k) for k in ref)
    print(f"{name:22s} worst rel grad diff {wg[0]:.3e} ({wg[1]})  worst param diff {wp[0]:.3e} ({wp[1]})")
