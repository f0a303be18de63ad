"""lt_mlp_backward_pair (csrc/lt_mlp.hip): the chain of input gradients of the actor and the critic - dz_l = (dz_{l+1} W_{l+1}) * ELU'(a_l),
autograd's `dz @ W` GEMMs and ELU-backward kernels of loss.backward() (loco_rl/loco_rl/algorithms/ppo.py:316) - as ONE launch of the
MFMA MLP kernel with the transposed weights, against the same chain in float64.  Tolerance: the results must be as close to f64 as
the f32 library path is (below 1e-6 of the largest |dz_l| of the layer; the weight gradients are compared with the library path's own error)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _nets(out0=12, out1=1, hidden=(512, 256, 128), d=348, seed=0, hidden1=None):
    import torch
    import torch.nn as nn

    torch.manual_seed(seed)
    def stack(out, hid):
        mods, prev = [], d
        for h in hid:
            mods += [nn.Linear(prev, h), nn.ELU()]
            prev = h
        mods.append(nn.Linear(prev, out))
        return nn.Sequential(*mods).cuda()
    return stack(out0, hidden), stack(out1, hidden1 or hidden)


def _ref_chain(seq, x, dy):
    """f64: (dz per hidden layer, dW per layer, db per layer)."""
    import torch
    import torch.nn as nn

    lin = [m for m in seq if isinstance(m, nn.Linear)]
    a, acts = x.double(), []
    for l in lin[:-1]:
        a = torch.nn.functional.elu(a @ l.weight.double().t() + l.bias.double())
        acts.append(a)
    g = dy.double()
    dz, dw, db = {}, {}, {}
    L = len(lin)
    for l in range(L - 1, -1, -1):
        inp = acts[l - 1] if l > 0 else x.double()
        if l < L - 1:
            g = g * torch.where(acts[l] > 0, torch.ones_like(acts[l]), acts[l] + 1.0)
            dz[l] = g
        dw[l], db[l] = g.t() @ inp, g.sum(0)
        g = g @ lin[l].weight.double()
    return dz, dw, db


@pytest.mark.parametrize("m,scale", [(24576, 1e-5), (6144, 1.0), (1000, 1e-8), (37, 1e-3)])
def test_fused_backward_matches_float64(m, scale):
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets()
    pair = M.PackedPair(actor, critic)
    g = torch.Generator(device="cuda").manual_seed(m)
    x0, x1 = torch.randn(m, 348, device="cuda", generator=g), torch.randn(m, 348, device="cuda", generator=g)
    dy0 = torch.randn(m, 12, device="cuda", generator=g) * scale
    dy0[::53] *= 200.0  # heavy-tailed rows, as PPO's are
    dy0[5] = 0.0
    dy1 = torch.randn(m, 1, device="cuda", generator=g) * scale
    res = {}
    for fused in (True, False):
        M.USE_FUSED_BACKWARD = fused
        try:
            grads = {p: torch.full_like(p, float("nan")) for net in (actor, critic) for p in net.parameters()}
            _, acts = pair.forward_raw(x0, x1)
            assert pair._fused_backward_ok(x0, x1, dy0, dy1) == fused
            pair.backward_raw(x0, x1, acts, dy0, dy1, grads)
            torch.cuda.synchronize()
            res[fused] = grads
            if fused:
                dzs, amaxs = pair._keep_bwd[:2]
                assert float(pair.saturated()) == 0.0 and pair.acts_split
                # the observation rows handed over in the split format (as PPO._direct_update does, once per update): same gradients
                again = {p: torch.full_like(p, float("nan")) for p in grads}
                pair.backward_raw(x0, x1, acts, dy0, dy1, again, (pair.split_rows(x0), pair.split_rows(x1)))
                torch.cuda.synchronize()
                for p in grads:
                    assert torch.equal(again[p], grads[p])
                # ... and with the producer's maxima of |dy| (lt_ppo_loss leaves them): ONE scale per network, every dz in the split
                # format - decoded here: (hi + lo / 64) / scale - and the same gradients to the f32-equivalent band
                third = {p: torch.full_like(p, float("nan")) for p in grads}
                pair.backward_raw(x0, x1, acts, dy0, dy1, third, (pair.split_rows(x0), pair.split_rows(x1)),
                                  dy_amax=(dy0.abs().max().reshape(1), dy1.abs().max().reshape(1)))
                torch.cuda.synchronize()
                sdz, _, _, _, scales = pair._keep_bwd
                split_dz = [[(t.view(torch.float16).view(t.shape[0], t.shape[1], 2).double() * torch.tensor([1.0, 1.0 / 64.0], device="cuda", dtype=torch.float64)).sum(-1)
                             / float(scales[k]) for t in sdz[k]] for k in range(2)]
                split_grads = third
                assert float(pair.saturated()) == 0.0
        finally:
            M.USE_FUSED_BACKWARD = True
    for k, (net, x, dy) in enumerate(((actor, x0, dy0), (critic, x1, dy1))):
        rdz, rdw, rdb = _ref_chain(net, x, dy)
        lin = [mm for mm in net if isinstance(mm, torch.nn.Linear)]
        for l, ref in rdz.items():
            top = float(ref.abs().max())
            err = float((dzs[k][l].double() - ref).abs().max())
            print(f"[mlp backward] m {m} net {k} dz_{l}: error / max {err / top:.2e}")
            assert err <= 1e-6 * top, (k, l, err, top)
            # rows are scaled one workgroup (16 .. 64 rows) at a time: a quiet block keeps its own precision next to a loud one
            pad = (-m) % 64
            e64 = torch.nn.functional.pad((dzs[k][l].double() - ref).abs(), (0, 0, 0, pad)).view(-1, 64 * ref.shape[1]).amax(1)
            t64 = torch.nn.functional.pad(ref.abs(), (0, 0, 0, pad)).view(-1, 64 * ref.shape[1]).amax(1)
            assert bool((e64 <= 2e-6 * t64).all()), (k, l, float((e64 / t64.clamp_min(1e-300)).max()))
            assert float(amaxs[k][l].max()) == pytest.approx(top, rel=1e-6)
        for l, ref in rdz.items():  # the split-format dz: one scale for the network, so the bound is relative to the layer's maximum
            assert float((split_dz[k][l] - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (k, l)
        for l in range(len(lin)):
            for name, got, ref in (("dW", split_grads[lin[l].weight], rdw[l]), ("db", split_grads[lin[l].bias], rdb[l])):
                top = float(ref.abs().max())
                eu = float((res[False][lin[l].weight if name == "dW" else lin[l].bias].double() - ref).abs().max())
                assert float((got.double() - ref).abs().max()) <= max(4.0 * eu, 1e-5 * top), (k, l, name, "split dz")
        for l in range(len(lin)):
            for name, got_f, got_u, ref in (("dW", res[True][lin[l].weight], res[False][lin[l].weight], rdw[l]), ("db", res[True][lin[l].bias], res[False][lin[l].bias], rdb[l])):
                top = float(ref.abs().max())
                ef, eu = float((got_f.double() - ref).abs().max()), float((got_u.double() - ref).abs().max())
                # (a column sum over 24 576 rows of mixed sign cancels to ~1 % of sum |dz|: 1e-5 of the result is 1e-7 of what was added)
                assert ef <= max(4.0 * eu, 1e-5 * top), (k, l, name, ef, eu, top)
    print(f"[mlp backward] m {m}: worst fused dW error / max |dW| "
          f"{max(float((res[True][p].double() - r).abs().max() / r.abs().max()) for net, x, dy in ((actor, x0, dy0), (critic, x1, dy1)) for p, r in zip([mm.weight for mm in net if isinstance(mm, torch.nn.Linear)], [_ref_chain(net, x, dy)[1][l] for l in range(4)])):.2e}")


def test_saturation_is_counted_not_silent():
    """Weights that amplify a gradient by > 500x through the chain push the scaled image past LT_MLP_INPUT_CLAMP: the counter says so."""
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(seed=3)
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(40.0)
    pair = M.PackedPair(actor, critic)
    m = 2048
    x = torch.randn(m, 348, device="cuda") * 0.01
    grads = {p: torch.zeros_like(p) for net in (actor, critic) for p in net.parameters()}
    _, acts = pair.forward_raw(x, x)
    pair.backward_raw(x, x, acts, torch.randn(m, 12, device="cuda"), torch.randn(m, 1, device="cuda"), grads)
    torch.cuda.synchronize()
    assert float(pair.saturated()) > 0


def test_shapes_outside_the_chain_kernel_take_the_library_path():
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(hidden=(64, 36))  # 36 is not a multiple of 8
    pair = M.PackedPair(actor, critic)
    x = torch.randn(256, 348, device="cuda")
    assert not pair._fused_backward_ok(x, x, torch.zeros(256, 12, device="cuda"), torch.zeros(256, 1, device="cuda"))
    lib, n = M._abi.load(), ctypes.c_size_t()
    assert lib.lt_mlp_backward_packed_floats(ctypes.byref(pair.a.desc), ctypes.byref(n)) != 0


@pytest.mark.parametrize("hidden,out0,out1,d,m", [((200, 136, 72), 12, 1, 348, 3000), ((64, 64), 8, 3, 100, 777), ((512, 8), 8, 1, 270, 9000), ((128,), 12, 1, 348, 2048)])
def test_fused_backward_on_other_network_shapes(hidden, out0, out1, d, m):
    """Other widths (multiples of 8 that are not multiples of 32, one hidden layer, the 270-wide rows of the locomotion task - padded to 272 for the first weight gradient -, other heads): fused backward pass ==
    float64 to the f32-equivalent band, split formats on every operand."""
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(out0=out0, out1=out1, hidden=hidden, d=d, seed=11)
    pair = M.PackedPair(actor, critic)
    g = torch.Generator(device="cuda").manual_seed(m)
    x0, x1 = torch.randn(m, d, device="cuda", generator=g), torch.randn(m, d, device="cuda", generator=g)
    dy0, dy1 = torch.randn(m, out0, device="cuda", generator=g) * 1e-4, torch.randn(m, out1, device="cuda", generator=g) * 1e-3
    grads = {p: torch.full_like(p, float("nan")) for net in (actor, critic) for p in net.parameters()}
    _, acts = pair.forward_raw(x0, x1)
    assert pair._fused_backward_ok(x0, x1, dy0, dy1) and pair.acts_split
    pair.backward_raw(x0, x1, acts, dy0, dy1, grads, (pair.split_rows(x0), pair.split_rows(x1)), dy_amax=(dy0.abs().max().reshape(1), dy1.abs().max().reshape(1)))
    torch.cuda.synchronize()
    assert float(pair.saturated()) == 0.0
    for net, x, dy in ((actor, x0, dy0), (critic, x1, dy1)):
        _, rdw, rdb = _ref_chain(net, x, dy)
        lin = [mm for mm in net if isinstance(mm, torch.nn.Linear)]
        for l in range(len(lin)):
            for got, ref in ((grads[lin[l].weight], rdw[l]), (grads[lin[l].bias], rdb[l])):
                assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-30, (hidden, l)


@pytest.mark.parametrize("m,hidden", [(24576, (512, 256, 128)), (6144, (512, 256, 128)), (1000, (200, 136, 72)), (16400, (200, 136, 72))])
def test_activations_in_the_split_format_decode_to_the_f32_activations(m, hidden):
    """lt_mlp_forward_pair(acts_split = 1): the dwords (f16 hi | f16 lo << 16) decode - hi + lo / 64 - to the activations the f32 form
    writes, to the format's 2^-21 (every row-tile variant, widths that are and are not multiples of 32); same outputs."""
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(hidden=hidden, seed=5)
    pair = M.PackedPair(actor, critic)
    g = torch.Generator(device="cuda").manual_seed(m)
    x0, x1 = torch.randn(m, 348, device="cuda", generator=g), 3.0 * torch.randn(m, 348, device="cuda", generator=g)
    (y0, y1), acts = pair.forward_raw(x0, x1, split=False)
    (z0, z1), sacts = pair.forward_raw(x0, x1, split=True)
    torch.cuda.synchronize()
    assert torch.equal(y0, z0) and torch.equal(y1, z1)
    for k in range(2):
        for a, s in zip(acts[k], sacts[k]):
            h = s.view(torch.float16).view(m, a.shape[1], 2).float()
            dec = h[..., 0] + h[..., 1] / 64.0
            assert float((dec - a).abs().max()) <= 2.0 ** -21 * max(1.0, float(a.abs().max())), (k, a.shape)


def test_one_launch_packing_equals_the_per_network_packs():
    """lt_mlp_pack_training (forward + transposed streams of both networks, one launch) writes the same bytes as lt_mlp_pack and
    lt_mlp_pack_backward per network (the buffers hold other weights' streams in between)."""
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(seed=9)
    pair = M.PackedPair(actor, critic)
    for net in (pair.a, pair.b):
        net.pack()
        net.pack_backward()
    torch.cuda.synchronize()
    want = [(net.packed.clone(), net.bpacked.clone()) for net in (pair.a, pair.b)]
    saved = [p.detach().clone() for net in (actor, critic) for p in net.parameters()]
    with torch.no_grad():
        for net in (actor, critic):
            for p in net.parameters():
                p.add_(torch.randn_like(p))
    for net in (pair.a, pair.b):  # other weights in the buffers
        net.pack()
        net.pack_backward()
    torch.cuda.synchronize()
    assert not torch.equal(pair.a.packed, want[0][0])
    with torch.no_grad():
        for p, v in zip([p for net in (actor, critic) for p in net.parameters()], saved):
            p.copy_(v)
    pair.pack_training(with_backward=True)
    torch.cuda.synchronize()
    for net, (p, bp) in zip((pair.a, pair.b), want):
        assert torch.equal(net.packed.view(torch.int32), p.view(torch.int32)) and torch.equal(net.bpacked.view(torch.int32), bp.view(torch.int32))


def test_actor_and_critic_of_different_depth_and_width():
    """The two stacks of a launch need not look alike (the reference's agent cfgs allow different actor / critic hidden dims)."""
    import torch

    from locotouch_amd.rl import mlp as M

    actor, critic = _nets(hidden=(256, 128), hidden1=(512, 256, 64, 32), seed=21)
    pair = M.PackedPair(actor, critic)
    m = 5000
    g = torch.Generator(device="cuda").manual_seed(1)
    x0, x1 = torch.randn(m, 348, device="cuda", generator=g), torch.randn(m, 348, device="cuda", generator=g)
    dy0, dy1 = torch.randn(m, 12, device="cuda", generator=g) * 1e-3, torch.randn(m, 1, device="cuda", generator=g) * 1e-5
    grads = {p: torch.full_like(p, float("nan")) for net in (actor, critic) for p in net.parameters()}
    _, acts = pair.forward_raw(x0, x1)
    assert pair._fused_backward_ok(x0, x1, dy0, dy1)
    pair.backward_raw(x0, x1, acts, dy0, dy1, grads, dy_amax=(dy0.abs().max().reshape(1), dy1.abs().max().reshape(1)))
    torch.cuda.synchronize()
    assert float(pair.saturated()) == 0.0
    for net, x, dy in ((actor, x0, dy0), (critic, x1, dy1)):
        _, rdw, rdb = _ref_chain(net, x, dy)
        lin = [mm for mm in net if isinstance(mm, torch.nn.Linear)]
        for l in range(len(lin)):
            for got, ref in ((grads[lin[l].weight], rdw[l]), (grads[lin[l].bias], rdb[l])):
                assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-30, l
