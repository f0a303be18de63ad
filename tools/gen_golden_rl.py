#!/usr/bin/env python3
"""Golden-vector generator for the trainer half of the path (`loco_rl`: GAE + PPO update).

BASELINE.json configs[0]: "loco_rl PPO update on a synthetic 64-env rollout buffer, CPU torch".
Runs ONLY in the build container: imports the reference's own `loco_rl` from /root/reference
(with the two in-memory stubs of SURVEY.md Appendix E for the absent `git` / `isaaclab` modules),
drives one seeded rollout + update, and stores the scalars/tensors our own trainer must reproduce.
The synthetic rollout itself is regenerated from the seed by the test (it is not stored).

    python tools/gen_golden_rl.py   # writes tests/golden/rl_ppo_cfg1.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/loco_rl")
sys.modules.setdefault("git", types.ModuleType("git"))
il, ilu = types.ModuleType("isaaclab"), types.ModuleType("isaaclab.utils")
ilu.configclass = lambda c: c
il.utils = ilu
sys.modules.update({"isaaclab": il, "isaaclab.utils": ilu})

import io  # noqa: E402
import contextlib  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from loco_rl.algorithms import PPO  # noqa: E402
from loco_rl.modules import ActorCritic  # noqa: E402

from tests.rl_synth import synth_rollout, PPO_CFG, POLICY_CFG, N_ENVS, N_STEPS, N_OBS, N_ACT  # noqa: E402


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    alg = PPO(ac, device="cpu", **PPO_CFG)
    alg.init_storage(N_ENVS, N_STEPS, [N_OBS], [N_OBS], [N_ACT])
    data = synth_rollout(seed=123)
    for t in range(N_STEPS):
        alg.act(data["obs"][t], data["critic_obs"][t])
        alg.process_env_step(data["rewards"][t], data["dones"][t], {"time_outs": data["time_outs"][t]})
    alg.compute_returns(data["last_critic_obs"])
    st = alg.storage
    out = dict(values=st.values.clone().numpy(), rewards=st.rewards.clone().numpy(),
               returns=st.returns.clone().numpy(), advantages=st.advantages.clone().numpy(),
               actions=st.actions.clone().numpy(), log_prob=st.actions_log_prob.clone().numpy())
    # minibatch order of the first epoch (generator draws one randperm), then the update itself
    rng_state = torch.get_rng_state()
    first_idx = torch.randperm(PPO_CFG["num_mini_batches"] * (N_ENVS * N_STEPS // PPO_CFG["num_mini_batches"]))
    torch.set_rng_state(rng_state)
    out["perm_head"] = first_idx[:64].numpy()
    v, s, e, _, _ = alg.update()
    out["losses"] = np.array([v, s, e], dtype=np.float64)
    out["learning_rate"] = np.array(alg.learning_rate)
    sd = alg.actor_critic.state_dict()
    out["param_names"] = np.array(list(sd.keys()))
    out["param_sum"] = np.array([p.double().sum().item() for p in sd.values()])
    out["param_abs_sum"] = np.array([p.double().abs().sum().item() for p in sd.values()])
    out["std"] = sd["std"].numpy()
    out["actor_last_bias"] = sd["actor.6.bias"].numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "rl_ppo_cfg1.npz"), **out)
    print("rl_ppo_cfg1.npz losses", out["losses"], "lr", out["learning_rate"])


if __name__ == "__main__":
    main()
