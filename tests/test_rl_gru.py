"""rl/gru.py against nn.GRU (same parameters): outputs, final state and every gradient."""
import pytest
import torch
import torch.nn as nn

from locotouch_amd.rl.gru import gru_sequence


def _check(device, L, B, I, H, tol):
    torch.manual_seed(0)
    gru = nn.GRU(I, H).to(device)
    x = torch.randn(L, B, I, device=device)
    h0 = 0.3 * torch.randn(1, B, H, device=device)
    g_out, g_h = torch.randn(L, B, H, device=device), torch.randn(1, B, H, device=device)
    res = []
    for fn in (lambda a, b: gru(a, b), lambda a, b: gru_sequence(gru, a, b)):
        gru.zero_grad()
        xa, ha = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        out, hn = fn(xa, ha)
        ((out * g_out).sum() + (hn * g_h).sum()).backward()
        res.append([out.detach(), hn.detach(), xa.grad, ha.grad] + [p.grad.clone() for p in gru.parameters()])
    names = ["out", "h_n", "dx", "dh0", "dW_ih", "dW_hh", "db_ih", "db_hh"]
    for n, a, b in zip(names, *res):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= tol * max(scale, 1.0), (n, float((a - b).abs().max()), scale)


def test_gru_sequence_matches_nn_gru_cpu():
    _check("cpu", 9, 5, 7, 12, 2e-5)
    _check("cpu", 40, 3, 64, 32, 5e-5)


@pytest.mark.gpu
def test_gru_sequence_matches_nn_gru_on_distillation_shape():
    _check("cuda:0", 500, 48, 64, 512, 3e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("L,B,I,H", [(20, 37, 64, 512), (7, 1, 5, 64), (33, 101, 64, 128), (3, 200, 16, 192), (2, 16, 8, 512), (1, 3, 8, 256)])
def test_hip_gru_kernels_on_ragged_shapes(L, B, I, H):
    """Row counts that are not multiples of the 16-row tile, a single row, other hidden sizes (multiples of 64), and the one- and
    two-step sequences where the backward recursion opens and closes at once (lt_gru_step_bwd_gates -> lt_gru_step_bwd_fused)."""
    import locotouch_amd.rl.gru as G

    assert G.use_hip_kernels
    _check("cuda:0", L, B, I, H, 2e-4)


@pytest.mark.gpu
def test_hidden_sizes_the_kernels_do_not_cover_take_the_torch_loop():
    _check("cuda:0", 6, 9, 8, 48, 2e-4)  # H = 48: not a multiple of 64 -> PyTorch-op time loop

