"""csrc/lt_wgrad.hip: the weight gradient dW = dz^T x of a Linear layer over a minibatch on the f16 matrix cores with the (hi, lo)
operand split - f32-equivalent: against an f64 reference its error must be no worse than an f32 GEMM's own (torch), at the
gradient magnitudes of the PPO loss (1e-4 ... 1e-9: the power-of-two scaling by max |dz| is what keeps them out of f16's
subnormals).  Shapes of the LocoTouch networks, a ragged M and the 348-wide first layer (tiles overhang the matrix)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _wgrad(dz, x, amax=None, x_split=False):
    import torch

    from locotouch_amd import _abi

    lib = _abi.load()
    vp = ctypes.c_void_p
    m, n = dz.shape
    k = x.shape[1]
    sp = int(lib.lt_wgrad_splits(m, n, k))
    slabs = torch.empty(int(lib.lt_wgrad_ws_floats(m, n, k)), device=dz.device)
    assert slabs.numel() == sp * n * k
    st = vp(torch.cuda.current_stream().cuda_stream)
    _abi.check(lib.lt_wgrad(vp(dz.data_ptr()), 0, vp(None), vp(x.data_ptr()), int(x_split), m, n, k, vp(amax.data_ptr()) if amax is not None else vp(None),
                            0 if amax is None else amax.numel(), vp(slabs.data_ptr()), vp(None), st), "lt_wgrad")
    return slabs.view(sp, n, k).sum(0), sp


@pytest.mark.parametrize("m,n,k", [(24576, 512, 348), (24576, 256, 512), (24576, 128, 256), (6144, 512, 348), (4100, 128, 64), (1000, 40, 52), (777, 4, 12)])
def test_split_f16_weight_gradient_is_f32_equivalent(m, n, k):
    import torch

    torch.manual_seed(m + n + k)
    x = torch.randn(m, k, device="cuda:0") * torch.rand(1, k, device="cuda:0") * 3.0  # columns of different scale, like observations
    # gradients as the PPO loss leaves them: ~1 / m, rows and columns of very different size, most of the mass in few entries
    dz = torch.randn(m, n, device="cuda:0") * torch.rand(m, 1, device="cuda:0") ** 4 * torch.rand(1, n, device="cuda:0") * (3.0 / m)
    amax = dz.abs().amax(dim=1)[: max(1, m // 48)].clone()
    amax[0] = dz.abs().max()
    got, sp = _wgrad(dz, x, amax)
    ref64 = dz.double().t() @ x.double()
    ref32 = dz.t() @ x
    scale = float(ref64.abs().max())
    err, err32 = float((got.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
    print(f"[wgrad] {m} x {n} x {k}: {sp} slices, max |dW| {scale:.3e}, error vs f64: split-f16 {err:.2e}, torch f32 GEMM {err32:.2e}")
    assert err < 2.0 * err32 + 2e-7 * scale, (err, err32, scale)
    # element by element, relative to the layer's largest gradient (Adam normalises per element, but an f32 GEMM offers no more)
    assert float((got.double() - ref64).abs().max()) < 2e-6 * scale


def test_unscaled_tiny_gradients_would_underflow_and_the_scale_fixes_it():
    import torch

    m, n, k = 6144, 128, 256
    torch.manual_seed(0)
    x = torch.randn(m, k, device="cuda:0")
    dz = torch.randn(m, n, device="cuda:0") * 1e-8  # below f16's smallest subnormal (6e-8): every hi half would round to zero
    ref = dz.double().t() @ x.double()
    plain, _ = _wgrad(dz, x, None)
    scaled, _ = _wgrad(dz, x, dz.abs().max().reshape(1))
    assert float((plain.double() - ref).abs().max()) > 0.01 * float(ref.abs().max())  # (f16 subnormals: a few bits are left)
    assert float((scaled.double() - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    # zero gradient: scale falls back to a finite power of two, the result is exactly zero
    zero, _ = _wgrad(torch.zeros_like(dz), x, torch.zeros(4, device="cuda:0"))
    assert float(zero.abs().max()) == 0.0


def test_backward_chain_with_the_split_f16_weight_gradients_matches_autograd():
    """rl/mlp.py backward_chain (what PPO._direct_update runs) against torch autograd on the same stack, in f64."""
    import torch

    from locotouch_amd.rl.mlp import PackedPair
    from locotouch_amd.rl.modules import build_mlp

    torch.manual_seed(3)
    m = 6144
    actor, critic = build_mlp(348, [512, 256, 128], 12, "elu").to("cuda:0"), build_mlp(348, [512, 256, 128], 1, "elu").to("cuda:0")
    pair = PackedPair(actor, critic)
    x0, x1 = torch.randn(m, 348, device="cuda:0"), torch.randn(m, 348, device="cuda:0")
    dy0, dy1 = torch.randn(m, 12, device="cuda:0") / m, torch.randn(m, 1, device="cuda:0") / m
    (y0, y1), acts = pair.forward_raw(x0, x1)
    grad_of = {p: torch.zeros_like(p) for net in (actor, critic) for p in net.parameters()}
    pair.backward_raw(x0, x1, acts, dy0, dy1, grad_of)
    for net, x, dy in ((actor, x0, dy0), (critic, x1, dy1)):
        ref = build_mlp(348, [512, 256, 128], dy.shape[1], "elu").to("cuda:0").double()
        ref.load_state_dict({k_: v.double() for k_, v in net.state_dict().items()})
        out = ref(x.double())
        out.backward(dy.double())
        for p, q in zip(net.parameters(), ref.parameters()):
            s = float(q.grad.abs().max())
            assert float((grad_of[p].double() - q.grad).abs().max()) < 3e-6 * s + 1e-12, (tuple(p.shape), s)


def test_partial_sums_launch_adds_every_job_in_a_fixed_order():
    """lt_partial_sums (csrc/lt_ppo.hip): several ordered sums of partials in one launch - the slabs of lt_wgrad, the per-block bias
    sums, the head kernels' [dW | db] blocks (count not a multiple of 4, outputs split over two tensors)."""
    import torch

    from locotouch_amd.rl.mlp import SumJobs

    g = torch.Generator(device="cuda").manual_seed(5)
    jobs, refs = SumJobs(), []
    for nblk, stride, count, split in ((21, 512 * 348, 512 * 348, 512 * 348), (256, 128 + 16, 129, 128), (1, 40, 37, 37), (513, 12 * 128 + 16, 12 * 128 + 12, 12 * 128),
                                       (7, 5, 5, 5), (128, 256, 256, 256)):
        ws = torch.randn(nblk * stride + 3, device="cuda", generator=g)[: nblk * stride]
        out0 = torch.full((split,), float("nan"), device="cuda")
        out1 = torch.full((count - split,), float("nan"), device="cuda") if count > split else None
        jobs.add(ws, nblk, stride, count, split, out0, out1)
        refs.append((ws.view(nblk, stride)[:, :count].double().sum(0), out0, out1, split, nblk))
    keep = list(jobs.jobs)
    jobs.launch()
    torch.cuda.synchronize()
    first = [(o0.clone(), None if o1 is None else o1.clone()) for _, o0, o1, _, _ in refs]
    for ref, o0, o1, split, nblk in refs:
        got = o0 if o1 is None else torch.cat((o0, o1))
        assert float((got.double() - ref).abs().max()) <= 2e-6 * max(1.0, nblk ** 0.5), nblk
    jobs.jobs = keep
    jobs.launch()
    torch.cuda.synchronize()
    for (a0, a1), (_, o0, o1, _, _) in zip(first, refs):
        assert torch.equal(a0, o0) and (a1 is None or torch.equal(a1, o1))


def test_operand_in_the_split_format_gives_the_same_gradient_as_the_clamped_rows():
    """lt_split_rows + lt_wgrad(x_split = 1): x handed over as (f16 hi | f16 lo << 16) dwords - what the training forward writes for
    its activations - equals the on-the-fly split of clamp(x, +-1000) bit for bit (same halves, same MFMAs)."""
    import torch

    from locotouch_amd import _abi

    lib, vp = _abi.load(), ctypes.c_void_p
    m, n, k = 6144, 256, 348
    g = torch.Generator(device="cuda").manual_seed(1)
    dz = torch.randn(m, n, device="cuda", generator=g) * 1e-4
    x = torch.randn(m, k, device="cuda", generator=g)
    x[::7, ::5] *= 900.0  # beyond the bound: saturated by the split format, as by the forward kernel
    am = dz.abs().max().reshape(1)
    xs = torch.empty(m, k, device="cuda", dtype=torch.int32)
    _abi.check(lib.lt_split_rows(vp(x.data_ptr()), vp(xs.data_ptr()), x.numel(), vp(torch.cuda.current_stream().cuda_stream)), "lt_split_rows")
    halves = xs.view(torch.float16).view(m, k, 2).float()
    rebuilt = halves[..., 0] + halves[..., 1] / 64.0
    xc = x.clamp(-1000.0, 1000.0)
    assert float((rebuilt - xc).abs().max()) <= 2.0 ** -21 * 1000.0 and float(((rebuilt - xc).abs() / xc.abs().clamp_min(1e-3)).max()) < 5e-7  # 2^-21
    a, _ = _wgrad(dz, xs.view(torch.float32), am, x_split=True)
    b, _ = _wgrad(dz, xc, am)
    assert torch.equal(a, b)
    assert lib.lt_split_rows(vp(x.data_ptr()), vp(xs.data_ptr()), 6, None) != 0


def test_tiled_form_gives_the_same_gradients(monkeypatch):
    """lt_wgrad128_kernel (128 x 128 tiles through LDS, LT_WGRAD_TILED=1; off by default - csrc/lt_wgrad.hip says why): same product,
    bias partials included, checked in a child process because the switch is read once per process."""
    import subprocess, sys, os

    code = r'''
import ctypes, torch
from locotouch_amd import _abi
lib, vp = _abi.load(), ctypes.c_void_p
for (m, n, k) in ((6144, 512, 348), (4100, 128, 256), (1000, 200, 100)):
    g = torch.Generator(device="cuda").manual_seed(m)
    dz = torch.randn(m, n, device="cuda", generator=g) * 1e-5
    x = torch.randn(m, k, device="cuda", generator=g)
    am = dz.abs().max().reshape(1)
    sp = int(lib.lt_wgrad_splits(m, n, k))
    slabs = torch.empty(sp * n * k + sp * n, device="cuda")
    st = vp(torch.cuda.current_stream().cuda_stream)
    _abi.check(lib.lt_wgrad(vp(dz.data_ptr()), 0, vp(None), vp(x.data_ptr()), 0, m, n, k, vp(am.data_ptr()), 1, vp(slabs.data_ptr()), vp(slabs[sp * n * k:].data_ptr()), st), "lt_wgrad")
    ref = dz.double().t() @ x.double()
    got = slabs[:sp * n * k].view(sp, n, k).double().sum(0)
    db = slabs[sp * n * k:].view(sp, n).double().sum(0)
    assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (m, n, k)
    assert float((db - dz.double().sum(0)).abs().max()) <= 2e-6 * float(dz.double().sum(0).abs().max()), (m, n, k)
print("ok")
'''
    env = dict(os.environ, LT_WGRAD_TILED="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_random_shapes_all_operand_formats_against_float64():
    """Twenty random (M, N, K) - ragged row counts, widths that leave partial tiles - in the three operand forms the update uses
    (f32 / f32, f32 / split x, split dz / split x), bias partials included."""
    import math
    import random

    import torch

    from locotouch_amd import _abi

    lib, vp = _abi.load(), ctypes.c_void_p
    rnd = random.Random(7)
    st = vp(torch.cuda.current_stream().cuda_stream)
    for case in range(20):
        m = rnd.choice([33, 64, 97, 500, 1000, 2049, 4100, 6144])
        n, k = 4 * rnd.randint(1, 128), 4 * rnd.randint(1, 128)
        g = torch.Generator(device="cuda").manual_seed(case)
        dz = torch.randn(m, n, device="cuda", generator=g) * 10.0 ** rnd.uniform(-8, -2)
        dz[::5] *= 50.0
        x = torch.randn(m, k, device="cuda", generator=g) * rnd.choice([0.1, 1.0, 30.0])
        ref, refb = dz.double().t() @ x.double(), dz.double().sum(0)
        am = dz.abs().max().reshape(1)
        xs, dzs = torch.empty_like(x), torch.empty_like(dz)
        sc = torch.tensor([2.0 ** (-math.floor(math.log2(float(am))))], device="cuda")
        _abi.check(lib.lt_split_rows(vp(x.data_ptr()), vp(xs.data_ptr()), x.numel(), st), "split x")
        _abi.check(lib.lt_split_rows(vp((dz * sc).data_ptr()), vp(dzs.data_ptr()), dz.numel(), st), "split dz")
        sp = int(lib.lt_wgrad_splits(m, n, k))
        for name, args in (("f32", (dz, 0, None, x, 0)), ("split x", (dz, 0, None, xs, 1)), ("split both", (dzs, 1, sc, xs, 1))):
            slabs = torch.full((sp * n * k + sp * n,), float("nan"), device="cuda")
            a_, asp, ascale, b_, bsp = args
            _abi.check(lib.lt_wgrad(vp(a_.data_ptr()), asp, vp(ascale.data_ptr()) if ascale is not None else vp(None), vp(b_.data_ptr()), bsp, m, n, k,
                                    vp(am.data_ptr()), 1, vp(slabs.data_ptr()), vp(slabs[sp * n * k:].data_ptr()), st), "lt_wgrad")
            got = slabs[:sp * n * k].view(sp, n, k).double().sum(0)
            db = slabs[sp * n * k:].view(sp, n).double().sum(0)
            assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-30, (case, name, m, n, k)
            assert float((db - refb).abs().max()) <= 2e-6 * float(dz.double().abs().sum(0).max()) + 1e-30, (case, name, m, n, k)
