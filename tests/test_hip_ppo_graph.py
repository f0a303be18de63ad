"""The sync-free PPO update on the GPU (KL -> learning-rate rule on a device scalar, statistics read once) against the
reference-shaped eager update with its host-side decisions."""
import pytest

pytestmark = pytest.mark.gpu


def _fill(alg, seed, n, T):
    import torch

    from tests.rl_synth import N_ACT, N_OBS

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


@pytest.mark.parametrize("m,clipped", [(4096, True), (1000, False), (1, True), (24576, True)])
def test_fused_loss_kernel_matches_the_torch_formulas(m, clipped):
    """csrc/lt_ppo.hip: loss terms, KL and the gradients w.r.t. mean, std and value against the reference's op chain
    (loco_rl/loco_rl/algorithms/ppo.py:251-311) written with torch ops in f64.  Tolerance: f32 sums over m rows."""
    import torch

    from locotouch_amd.rl.fused_loss import fused_ppo_loss

    A, clip, vcoef, ecoef = 12, 0.2, 1.0, 0.01
    g = torch.Generator(device="cuda:0").manual_seed(m)
    r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)  # noqa: E731
    old_mu, old_sigma = r(m, A), (0.5 + 0.3 * torch.rand(m, A, device="cuda:0", generator=g))
    actions = old_mu + old_sigma * r(m, A)
    old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(actions).sum(-1, keepdim=True)
    mu0, std0, value0 = old_mu + 0.15 * r(m, A), (0.5 + 0.3 * torch.rand(A, device="cuda:0", generator=g)), r(m, 1)
    adv, returns, old_values = r(m, 1), r(m, 1), value0 + 0.3 * r(m, 1)

    def leaves(dt):
        return [t.detach().to(dt).requires_grad_(True) for t in (mu0, std0, value0)]

    mu, std, value = leaves(torch.float32)
    loss, surr, vl, ent, kl = fused_ppo_loss(mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped)
    (2.0 * loss).backward()  # a non-unit upstream gradient

    mu_d, std_d, value_d = leaves(torch.float64)
    d = lambda t: t.double()  # noqa: E731
    dist = torch.distributions.Normal(mu_d, std_d.expand_as(mu_d))
    logp = dist.log_prob(d(actions)).sum(-1)
    sig = std_d.expand_as(mu_d)
    kl_ref = torch.sum(torch.log(sig / d(old_sigma) + 1e-5) + (d(old_sigma) ** 2 + (d(old_mu) - mu_d) ** 2) / (2 * sig ** 2) - 0.5, -1).mean()
    ratio = torch.exp(logp - d(old_logp).squeeze(-1))
    a_ = d(adv).squeeze(-1)
    surr_ref = torch.max(-a_ * ratio, -a_ * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
    if clipped:
        vclip = d(old_values) + (value_d - d(old_values)).clamp(-clip, clip)
        vl_ref = torch.max((value_d - d(returns)) ** 2, (vclip - d(returns)) ** 2).mean()
    else:
        vl_ref = ((d(returns) - value_d) ** 2).mean()
    ent_ref = dist.entropy().sum(-1).mean()
    loss_ref = surr_ref + vcoef * vl_ref - ecoef * ent_ref
    (2.0 * loss_ref).backward()
    for got, ref in ((loss, loss_ref), (surr, surr_ref), (vl, vl_ref), (ent, ent_ref), (kl, kl_ref)):
        assert abs(float(got) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    torch.testing.assert_close(mu.grad.double(), mu_d.grad, rtol=2e-4, atol=2e-6 / m)
    torch.testing.assert_close(value.grad.double(), value_d.grad, rtol=2e-4, atol=2e-6 / m)
    torch.testing.assert_close(std.grad.double(), std_d.grad, rtol=5e-4, atol=2e-5)


def test_update_with_fused_loss_equals_update_with_torch_ops():
    import torch

    from locotouch_amd.rl import PPO, ActorCritic
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

    n, T = 512, 24
    cfg = dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=4)
    algs = []
    for fused in (True, False):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", fused_loss=fused, **cfg)
        alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
        algs.append(alg)
    a, b = algs
    assert a.fused_loss and not b.fused_loss
    lrs = set()
    for it in range(4):
        outs = []
        for alg in (a, b):
            _fill(alg, 100 + it, n, T)
            torch.manual_seed(7 + it)
            outs.append(alg.update())
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert abs(x - y) <= 2e-4 * max(1.0, abs(y)), (it, outs)
        assert abs(a.learning_rate - b.learning_rate) <= 1e-9 + 1e-6 * b.learning_rate
        lrs.add(round(a.learning_rate, 9))
    assert len(lrs) > 1
    # 32 Adam steps: where a gradient component is ~0 its sign is rounding noise and Adam turns that into +-lr per step, so a
    # few elements in a thousand sit a couple of learning rates apart (the kernel's own gradients are pinned in f64 above)
    for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()):
        torch.testing.assert_close(pa, pb, rtol=5e-3, atol=4e-3)
        assert float((pa - pb).abs().mean()) < 3e-4


def test_device_side_update_equals_host_side_update():
    import torch

    from locotouch_amd.rl import PPO, ActorCritic
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

    n, T = 512, 24
    cfg = dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=4)

    def make(graph):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", device_update=graph, fused_loss=False, **cfg)
        alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
        return alg

    def fill(alg, seed):
        _fill(alg, seed, n, T)

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
