"""rl/linear.py: the split-K weight-gradient path gives nn.Linear's gradients (summation order aside)."""
import pytest
import torch
import torch.nn as nn

from locotouch_amd.rl.linear import Linear


def _check(device, m, k, n, rtol):
    torch.manual_seed(0)
    ref = nn.Linear(k, n).to(device)
    lin = Linear(k, n).to(device)
    lin.load_state_dict(ref.state_dict())
    assert list(lin.state_dict()) == list(ref.state_dict())
    x = torch.randn(m, k, device=device)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(m, n, device=device)
    y1, y2 = ref(x1), lin(x2)
    assert torch.equal(y1, y2)
    y1.backward(g), y2.backward(g)
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=rtol, atol=rtol * float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(lin.bias.grad, ref.bias.grad, rtol=rtol, atol=rtol * float(ref.bias.grad.abs().max()))
    torch.testing.assert_close(x2.grad, x1.grad, rtol=rtol, atol=rtol * float(x1.grad.abs().max()))


def test_split_k_linear_matches_nn_linear_cpu(monkeypatch):
    monkeypatch.setattr(Linear, "split_k_min_rows", 64)
    _check("cpu", 256, 37, 19, 2e-5)
    # below the threshold / not divisible: the stock path, bit-identical
    lin, ref = Linear(8, 4), nn.Linear(8, 4)
    lin.load_state_dict(ref.state_dict())
    x = torch.randn(30, 8)
    lin(x).sum().backward(), ref(x).sum().backward()
    assert torch.equal(lin.weight.grad, ref.weight.grad)


@pytest.mark.gpu
def test_split_k_linear_matches_nn_linear_on_update_shapes():
    for k, n in [(348, 512), (512, 256), (256, 128), (128, 12), (128, 1)]:
        _check("cuda:0", 24576, k, n, 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(4099, 128, 12), (5000, 36, 5), (4096, 1024, 16), (4100, 1024, 8), (24576, 256, 3), (4097, 4, 1), (6000, 128, 14)])
def test_narrow_head_gradients_on_ragged_shapes(m, k, n):
    """csrc/lt_ppo.hip lt_head_wgrad (n <= 16 outputs): weight + bias gradient in one pass; 13 .. 16 outputs use 64 KiB + of LDS partials
    (raised dynamic-LDS limit)."""
    _check("cuda:0", m, k, n, 3e-5)
