"""The sync-free PPO update on the GPU (KL -> learning-rate rule on a device scalar, statistics read once) against the
reference-shaped eager update with its host-side decisions."""
import pytest

pytestmark = pytest.mark.gpu


def test_device_side_update_equals_host_side_update():
    import torch

    from locotouch_amd.rl import PPO, ActorCritic
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

    n, T = 512, 24
    cfg = dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=4)

    def make(graph):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", device_update=graph, **cfg)
        alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
        return alg

    def fill(alg, seed):
        g = torch.Generator(device="cuda:0").manual_seed(seed)
        ac = alg.actor_critic
        with torch.no_grad():
            for t in range(T):
                o = torch.randn(n, N_OBS, device="cuda:0", generator=g)
                c = o + 0.01 * torch.randn(n, N_OBS, device="cuda:0", generator=g)
                ac.update_distribution(o)
                act = ac.action_mean + ac.action_std * torch.randn(n, N_ACT, device="cuda:0", generator=g)
                alg._t = dict(actions=act, values=ac.evaluate(c), log_prob=ac.get_actions_log_prob(act), mu=ac.action_mean.clone(),
                              sigma=ac.action_std.clone(), obs=o, critic_obs=c)
                alg.process_env_step(torch.randn(n, device="cuda:0", generator=g),
                                     (torch.rand(n, device="cuda:0", generator=g) < 0.05).long(), {})
            alg.compute_returns(torch.randn(n, N_OBS, device="cuda:0", generator=g))

    a, b = make(True), make(False)
    lrs = []
    for it in range(4):
        outs = []
        for alg in (a, b):
            fill(alg, 100 + it)
            torch.manual_seed(7 + it)  # same minibatch permutation on both sides
            outs.append(alg.update())
        lrs.append((a.learning_rate, b.learning_rate))
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert abs(x - y) <= 2e-4 * max(1.0, abs(y)), (it, outs)
        assert abs(a.learning_rate - b.learning_rate) <= 1e-9 + 1e-6 * b.learning_rate, lrs
    assert a._lr_t is not None and b._lr_t is None, "side a must really have run the device-side path"
    assert len({round(x, 9) for x, _ in lrs}) > 1, "the adaptive rule must have moved the learning rate"
    for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()):
        torch.testing.assert_close(pa, pb, rtol=5e-3, atol=5e-4)
