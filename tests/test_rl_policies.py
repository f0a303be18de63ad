"""ActorCriticRecurrent (GRU) and ActorCriticEncoder + the PPO recurrent branch against tests/golden/rl_policies.npz, recorded from
the reference's own `loco_rl` classes on the seeded rollout of tests/rl_synth.py::policy_case (tools/gen_golden_rl_policies.py):
parameter names / shapes / seeded initial weights, the rollout (hidden state carried, reset at dones, saved per step),
advantages, the three losses of one update, the adaptive learning rate and the post-update parameters.  CPU only."""
import os

import numpy as np
import pytest
import torch

from locotouch_amd.rl import PPO, ActorCriticEncoder, ActorCriticRecurrent
from tests.rl_synth import policy_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rl_policies.npz")


@pytest.mark.parametrize("kind", ["recurrent", "encoder"])
def test_policy_class_rollout_and_update_match_reference(kind):
    torch.set_num_threads(1)
    g = np.load(GOLD)
    case = policy_case(kind)
    torch.manual_seed(case["seed"])
    ac = (ActorCriticRecurrent if kind == "recurrent" else ActorCriticEncoder)(*case["args"], **case["kwargs"])
    sd = ac.state_dict()
    assert list(sd.keys()) == g[f"{kind}_keys"].tolist()
    assert [str(tuple(v.shape)) for v in sd.values()] == g[f"{kind}_shapes"].tolist()
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g[f"{kind}_init_sums"], rtol=1e-9, atol=1e-9)
    alg = PPO(ac, device="cpu", **case["ppo"])
    N, T, D, A = case["N"], case["T"], case["D"], case["A"]
    alg.init_storage(N, T, [D], [D], [A])
    torch.manual_seed(case["seed"] + 1)
    with torch.inference_mode():
        for t in range(T):
            alg.act(case["obs"][t], case["cobs"][t])
            np.testing.assert_allclose(alg._t["log_prob"].numpy(), g[f"{kind}_logp"][t], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(alg._t["values"].numpy(), g[f"{kind}_values"][t], rtol=1e-5, atol=1e-6)
            alg.process_env_step(case["rewards"][t], case["dones"][t], {})
        alg.compute_returns(case["last_cobs"])
    np.testing.assert_allclose(alg.storage.actions.numpy(), g[f"{kind}_actions"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alg.storage.advantages.numpy(), g[f"{kind}_adv"], rtol=1e-4, atol=1e-5)
    if kind == "recurrent":
        assert alg.storage.saved_hidden_states_a[0].shape == (T, 1, N, 24) and alg.storage.saved_hidden_states_c[0].shape == (T, 1, N, 24)
        assert (alg.storage.saved_hidden_states_a[0][0] == 0).all()  # before the first step the memories hold nothing
        d = case["dones"].bool()
        t, e = [(int(a), int(b)) for a, b in d.nonzero() if a < T - 1][0]
        assert (alg.storage.saved_hidden_states_a[0][t + 1, 0, e] == 0).all()  # reset right after a done
    torch.manual_seed(case["seed"] + 2)
    losses = alg.update()
    np.testing.assert_allclose(losses[:3], g[f"{kind}_losses"], rtol=2e-5, atol=1e-6)
    assert abs(alg.learning_rate - float(g[f"{kind}_lr"])) < 1e-12
    post = ac.state_dict()
    np.testing.assert_allclose([float(v.double().sum()) for v in post.values()], g[f"{kind}_post_sums"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in post.values()], g[f"{kind}_post_abs"], rtol=1e-4, atol=2e-4)


def test_runner_dispatches_the_policy_class():
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.rl import OnPolicyRunner
    from tests.distill_synth import ScriptedEnv

    env = ScriptedEnv(6, form="tuple")
    env.num_obs = 348
    cfg = train_cfg("Isaac-RandCylinderTransportTeacher-LocoTouch-v1")
    cfg["policy"] = dict(class_name="ActorCriticRecurrent", init_noise_std=1.0, actor_hidden_dims=[32], critic_hidden_dims=[32],
                         activation="elu", rnn_type="gru", rnn_hidden_size=16, rnn_num_layers=1)
    cfg["num_steps_per_env"], cfg["algorithm"]["num_mini_batches"], cfg["algorithm"]["num_learning_epochs"] = 6, 2, 1
    r = OnPolicyRunner(env, cfg, log_dir=None, device="cpu")
    assert type(r.alg.actor_critic).__name__ == "ActorCriticRecurrent"
    r.learn(2)
    assert len(r.history) == 2 and np.isfinite(r.history[-1]["Loss/value_function"])
    cfg["policy"]["class_name"] = "ActorCriticRnnEncoder"
    with pytest.raises(NotImplementedError, match="ActorCriticRnnEncoder"):
        OnPolicyRunner(env, cfg, log_dir=None, device="cpu")
