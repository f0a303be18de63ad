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
    # index form: the batch tensors are a larger storage, the minibatch is rows idx of it
    perm = torch.randperm(3 * m, device="cuda:0", generator=g)
    idx, big = perm[:m], lambda t: torch.randn(3 * m, *t.shape[1:], device="cuda:0", generator=g).index_copy_(0, perm[:m], t)  # noqa: E731
    mu_i, std_i, value_i = leaves(torch.float32)
    big_sigma = (torch.rand(3 * m, A, device="cuda:0", generator=g) + 0.1).index_copy_(0, idx, old_sigma)
    out_i = fused_ppo_loss(mu_i, std_i, value_i, big(actions), big(old_logp), big(adv), big(returns), big(old_values), big(old_mu),
                           big_sigma, clip, vcoef, ecoef, clipped, idx=idx)
    (2.0 * out_i[0]).backward()
    torch.testing.assert_close(torch.stack(out_i), torch.stack((loss, surr, vl, ent, kl)), rtol=1e-5, atol=1e-6)
    assert torch.equal(mu_i.grad, mu.grad) and torch.equal(value_i.grad, value.grad)

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
        assert abs(float(got.detach()) - float(ref.detach())) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    torch.testing.assert_close(mu.grad.double(), mu_d.grad, rtol=2e-4, atol=2e-6 / m)
    torch.testing.assert_close(value.grad.double(), value_d.grad, rtol=2e-4, atol=2e-6 / m)
    torch.testing.assert_close(std.grad.double(), std_d.grad, rtol=5e-4, atol=2e-5)


def test_update_with_fused_loss_equals_update_with_torch_ops():
    import torch

    from locotouch_amd.rl import PPO, ActorCritic
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

    n, T = 1024, 24  # 6144-row minibatches: the fused Linear -> ELU nodes and split-K weight gradients are on the path
    cfg = dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=4)
    algs = []
    for fused in (True, False):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", fused_loss=fused, fused_adam=fused, **cfg)
        alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
        algs.append(alg)
    a, b = algs
    assert a.fused_loss and a._flat_adam is not None and not b.fused_loss and b._flat_adam is None
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


def test_direct_update_without_host_reads_equals_the_autograd_form():
    """PPO._direct_update (no autograd graph, the KL -> learning-rate rule and the statistics on the device, gradients written into
    the flat bucket in place, ONE host read per update) against PPO._fused_update (same forward / loss kernels behind autograd nodes,
    the host deciding the learning rate from the KL of every minibatch step, loco_rl/loco_rl/algorithms/ppo.py:273-281).

    What can be pinned tightly is ONE optimizer step from identical parameters: the two forms differ only in who adds the partial
    sums and in the weight-gradient kernel (csrc/lt_wgrad.hip there, ~1e-8 of max |dW| from f64; the library's f32 GEMM here, ~2e-7).
    Over several steps PPO itself is a discontinuous map of its parameters (the ratio clip and the value clip pick a branch per
    sample, and a heavy-tailed sample that changes its branch moves the minibatch gradient by percents), so there the check is
    statistical: same learning-rate decisions, same losses, parameters within a few learning rates."""
    import torch

    from locotouch_amd.rl import PPO, ActorCritic, tuned_gemms
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

    tuned_gemms.disable()  # both forms on the library's default GEMM algorithms (TunableOp's state does not reach the autograd thread)
    n, T = 1024, 24

    def pair_of(cfg):
        out = []
        for direct in (True, False):
            torch.manual_seed(0)
            alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", direct_update=direct, **cfg)
            alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
            out.append(alg)
        assert out[0].direct_update and not out[1].direct_update and out[0]._flat_adam is not None and out[1]._flat_adam is not None
        return out

    # (1) one step from identical parameters: gradients to the f32 GEMM's own error, parameters to one Adam step of it
    a, b = pair_of(dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=1, tuned_gemms=False))
    outs = []
    for alg in (a, b):
        _fill(alg, 200, n, T)
        torch.manual_seed(11)
        outs.append(alg.update())
    for x, y in zip(outs[0][:3], outs[1][:3]):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(y)), outs
    for (name, pa), pb in zip(a.actor_critic.named_parameters(), b.actor_critic.parameters()):
        assert float((pa.grad - pb.grad).abs().max()) <= 3e-6 * float(pb.grad.abs().max()), name
        assert pa.grad.data_ptr() >= a._flat_adam.flat_g.data_ptr()
        # Adam's first step is lr * g / (|g| + eps): where |g| ~ eps = 1e-8 an error of 1e-9 is a visible fraction of lr
        torch.testing.assert_close(pa, pb, rtol=0, atol=2 * a.learning_rate)
        assert float((pa - pb).abs().mean()) < 1e-6, name

    # (2) four iterations of four minibatch steps
    a, b = pair_of(dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=4, tuned_gemms=False))
    lrs = set()
    for it in range(4):
        outs = []
        for alg in (a, b):
            _fill(alg, 200 + it, n, T)
            torch.manual_seed(11 + it)
            outs.append(alg.update())
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (it, outs)
        assert abs(a.learning_rate - b.learning_rate) <= 1e-5 * b.learning_rate  # (repeated x / 1.5 in f32 against the host's f64)
        assert a.optimizer.param_groups[0]["lr"] == a.learning_rate
        lrs.add(round(a.learning_rate, 9))
        for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()):
            torch.testing.assert_close(pa, pb, rtol=0, atol=4e-3)
            assert float((pa - pb).abs().mean()) < 1e-4 * (it + 1)
    assert len(lrs) > 1, "the adaptive rule must have moved the learning rate"
    # fixed schedule: the rule kernel leaves the rate alone
    cfg = dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=4, tuned_gemms=False)
    torch.manual_seed(0)
    c = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cuda:0", **dict(cfg, schedule="fixed"))
    c.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
    _fill(c, 300, n, T)
    c.update()
    assert c.learning_rate == pytest.approx(cfg["learning_rate"], rel=1e-7)


@pytest.mark.parametrize("m,n", [(24576, 512), (4100, 256), (5000, 128), (4096, 400), (4097, 4)])
def test_linear_elu_node_matches_torch(m, n):
    """`MLPSequential`'s fused Linear -> ELU node (csrc/lt_ppo.hip lt_elu_backward_bias) against nn.Linear + nn.ELU."""
    import torch

    from locotouch_amd.rl.linear import Linear, MLPSequential

    torch.manual_seed(m + n)
    k = 96
    net = MLPSequential(Linear(k, n), torch.nn.ELU(), Linear(n, 8)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(k, n), torch.nn.ELU(), torch.nn.Linear(n, 8)).cuda()
    ref.load_state_dict(net.state_dict())
    x = torch.randn(m, k, device="cuda:0")
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    w = torch.randn(m, 8, device="cuda:0") / m
    y1, y2 = net(x1), ref(x2)
    assert y1.grad_fn is not None and "LinearELU" in type(y1.grad_fn.next_functions[0][0]).__name__, "the fused node must have been taken"
    torch.testing.assert_close(y1, y2, rtol=1e-5, atol=1e-5)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-4, atol=1e-8)
    for (name, p1), p2 in zip(net.named_parameters(), ref.parameters()):
        torch.testing.assert_close(p1.grad, p2.grad, rtol=2e-4, atol=2e-6, msg=lambda s_: f"{name}: {s_}")


def test_flat_adam_matches_clip_grad_norm_plus_torch_adam():
    """rl/flat_adam.py (csrc/lt_ppo.hip lt_adam_clip_step) against clip_grad_norm_ + torch.optim.Adam.step(), including the views
    kept in the optimizer state, a state_dict round trip through a second optimizer, and steps where the clip does / does not bind."""
    import copy

    import torch

    from locotouch_amd.rl import ActorCritic
    from locotouch_amd.rl.flat_adam import FlatAdam
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG

    torch.manual_seed(3)
    a = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG).cuda()
    b = copy.deepcopy(a)
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-3), torch.optim.Adam(b.parameters(), lr=1e-3)
    fa = FlatAdam(oa)
    assert all(p.data_ptr() >= fa.flat_p.data_ptr() for p in a.parameters()) and fa.n % 64 == 0
    g = torch.Generator(device="cuda:0").manual_seed(0)
    for it in range(6):
        scale = 10.0 if it % 2 else 1e-3  # norm above / below max_norm = 1
        fa.zero_grad()
        for i, (pa, pb) in enumerate(zip(a.parameters(), b.parameters())):
            gr = scale * torch.randn(pb.shape, device="cuda:0", generator=g)
            pb.grad = gr.clone()
            if not (it == 4 and i == 2):  # one step leaves a parameter without a gradient: counts as zero
                pa.grad = gr.clone()
            else:
                pb.grad.zero_()
        if it == 3:  # learning-rate change + a checkpoint round trip of the optimizer state in between
            for grp in (*oa.param_groups, *ob.param_groups):
                grp["lr"] = 4e-4
            sd = copy.deepcopy(oa.state_dict())
            oa.load_state_dict(sd)  # re-binds the state tensors: must be re-adopted, not lost
        norm_b = torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        ob.step()
        fa.step(1.0)
        torch.testing.assert_close(fa.grad_norm[0], norm_b, rtol=1e-5, atol=0)
        for pa, pb in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(pa.grad, pb.grad, rtol=1e-5, atol=1e-9)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert sa.keys() == sb.keys()
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 6.0
        torch.testing.assert_close(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    assert (fa.flat_p.view(-1, 64)[:, :].abs().sum() > 0) and float(fa.flat_m.abs().sum()) > 0


@pytest.mark.parametrize("m", [24576, 4100, 33])
def test_packed_pair_training_forward_and_backward_match_the_torch_modules(m):
    """rl/mlp.py PackedPair (csrc/lt_mlp.hip lt_mlp_forward_pair + the hand-written backward chain): outputs and every parameter
    gradient of actor and critic against the nn.Sequential stacks themselves.  f32-equivalent arithmetic (split-fp16 MFMA with error
    compensation): tolerances are those of an f32 GEMM."""
    import torch

    from locotouch_amd.rl import ActorCritic
    from locotouch_amd.rl.mlp import PackedPair
    from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG

    torch.manual_seed(m)
    ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG).cuda()
    pair = PackedPair(ac.actor, ac.critic)
    x0, x1 = torch.randn(m, N_OBS, device="cuda:0"), torch.randn(m, N_OBS, device="cuda:0")
    g0, g1 = torch.randn(m, N_ACT, device="cuda:0") / m, torch.randn(m, 1, device="cuda:0") / m
    res = []
    for fn in (lambda: (ac.actor(x0), ac.critic(x1)), lambda: pair(x0, x1)):
        ac.zero_grad()
        mu, v = fn()
        ((mu * g0).sum() + (v * g1).sum()).backward()
        res.append([mu.detach(), v.detach()] + [p.grad.clone() for n_, p in ac.named_parameters() if n_ != "std"])
    names = ["mu", "value"] + [n_ for n_, _ in ac.named_parameters() if n_ != "std"]
    for name, a, b in zip(names, *res):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * max(scale, 1e-6) + 1e-7, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("n,T", [(4096, 24), (405, 7), (1, 3)])
def test_gae_kernel_matches_the_tensor_op_recursion(n, T):
    """csrc/lt_ppo.hip lt_gae against RolloutStorage.compute_returns' per-step tensor ops (the CPU path, pinned to the reference's
    golden in tests/test_rl_parity.py), incl. the advantage normalisation that follows."""
    import torch

    from locotouch_amd.rl.storage import RolloutStorage

    g = torch.Generator().manual_seed(n + T)
    cpu, gpu = RolloutStorage(n, T, 4, 4, 2, device="cpu"), RolloutStorage(n, T, 4, 4, 2, device="cuda:0")
    for name in ("rewards", "values"):
        v = torch.randn(T, n, 1, generator=g)
        getattr(cpu, name).copy_(v), getattr(gpu, name).copy_(v)
    d = (torch.rand(T, n, 1, generator=g) < 0.1).to(torch.uint8)
    cpu.dones.copy_(d), gpu.dones.copy_(d)
    last = torch.randn(n, 1, generator=g)
    for norm in (False, True):
        cpu.compute_returns(last, 0.99, 0.95, normalize_advantage=norm)
        gpu.compute_returns(last.cuda(), 0.99, 0.95, normalize_advantage=norm)
        torch.testing.assert_close(gpu.returns.cpu(), cpu.returns, rtol=1e-5, atol=1e-5)
        if n * T > 1 or not norm:
            torch.testing.assert_close(gpu.advantages.cpu(), cpu.advantages, rtol=1e-4, atol=1e-5)


def test_direct_update_on_shapes_outside_the_fused_backward_takes_the_library_path():
    """Hidden widths that are not multiples of 8 (36) have no chain kernel (lt_mlp_backward_packed_floats refuses): the direct update
    then keeps f32 activations and runs `dz @ W` / the weight gradients as library GEMMs - same results as the autograd form."""
    import torch

    from locotouch_amd.rl import PPO, ActorCritic, tuned_gemms
    from tests.rl_synth import N_ACT, N_OBS, PPO_CFG

    tuned_gemms.disable()
    pol = dict(init_noise_std=1.0, actor_hidden_dims=[64, 36], critic_hidden_dims=[64, 36], activation="elu")
    n, T = 256, 24
    algs = []
    for direct in (True, False):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **pol), device="cuda:0", direct_update=direct,
                  **dict(PPO_CFG, num_learning_epochs=1, num_mini_batches=1, tuned_gemms=False))
        alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
        algs.append(alg)
    outs = []
    for alg in algs:
        _fill(alg, 400, n, T)
        torch.manual_seed(3)
        outs.append(alg.update())
    a, b = algs
    from locotouch_amd.rl.mlp import PackedPair

    assert not PackedPair(a.actor_critic.actor, a.actor_critic.critic)._fused_backward_possible(torch.zeros(8, N_OBS, device="cuda"), torch.zeros(8, N_OBS, device="cuda"))
    for x, y in zip(outs[0][:3], outs[1][:3]):
        assert abs(x - y) <= 1e-5 * max(1.0, abs(y))
    for pa, pb in zip(a.actor_critic.parameters(), b.actor_critic.parameters()):
        assert float((pa.grad - pb.grad).abs().max()) <= 3e-6 * float(pb.grad.abs().max()) + 1e-12
