"""Trainer parity (BASELINE.json configs[0]): locotouch_amd.rl against what the reference's own loco_rl produced on the
same seeded 64-env x 24-step synthetic rollout (tests/golden/rl_ppo_cfg1.npz, tools/gen_golden_rl.py).  CPU torch."""
import os

import numpy as np
import torch

from locotouch_amd.rl import PPO, ActorCritic
from tests.rl_synth import N_ACT, N_ENVS, N_OBS, N_STEPS, POLICY_CFG, PPO_CFG, synth_rollout

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rl_ppo_cfg1.npz")


def _run():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    alg = PPO(ac, device="cpu", **PPO_CFG)
    alg.init_storage(N_ENVS, N_STEPS, [N_OBS], [N_OBS], [N_ACT])
    data = synth_rollout(seed=123)
    for t in range(N_STEPS):
        alg.act(data["obs"][t], data["critic_obs"][t])
        alg.process_env_step(data["rewards"][t], data["dones"][t], {"time_outs": data["time_outs"][t]})
    alg.compute_returns(data["last_critic_obs"])
    return alg


def test_gae_and_rollout_match_reference():
    g = np.load(GOLD)
    alg = _run()
    st = alg.storage
    # same seed, same construction order, same sampling call -> identical actions / log-probs / values
    np.testing.assert_allclose(st.actions.numpy(), g["actions"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st.values.numpy(), g["values"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st.actions_log_prob.numpy(), g["log_prob"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(st.rewards.numpy(), g["rewards"], rtol=0, atol=1e-6)  # incl. time-out bootstrap
    np.testing.assert_allclose(st.returns.numpy(), g["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(st.advantages.numpy(), g["advantages"], rtol=1e-5, atol=1e-5)


def test_ppo_update_matches_reference():
    g = np.load(GOLD)
    alg = _run()
    state = torch.get_rng_state()
    perm = torch.randperm(PPO_CFG["num_mini_batches"] * (N_ENVS * N_STEPS // PPO_CFG["num_mini_batches"]))
    torch.set_rng_state(state)
    np.testing.assert_array_equal(perm[:64].numpy(), g["perm_head"])  # the one randperm of the update
    v, s, e, _, _ = alg.update()
    np.testing.assert_allclose([v, s, e], g["losses"], rtol=1e-5, atol=1e-6)
    assert abs(alg.learning_rate - float(g["learning_rate"])) < 1e-12
    sd = alg.actor_critic.state_dict()
    assert list(sd.keys()) == list(g["param_names"])  # checkpoint key set / order
    np.testing.assert_allclose([p.double().sum().item() for p in sd.values()], g["param_sum"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose([p.double().abs().sum().item() for p in sd.values()], g["param_abs_sum"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sd["std"].numpy(), g["std"], rtol=1e-5, atol=1e-6)
    assert sum(p.numel() for p in alg.actor_critic.parameters()) == 687513
