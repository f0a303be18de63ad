"""rl/models.py::Conv2dAsGemm (the layer's matrix from its response to the identity basis, then one GEMM) against nn.Conv2d."""
import pytest
import torch
import torch.nn as nn

from locotouch_amd.rl.models import CNN2dHead, Conv2dAsGemm


def _check(device, n):
    torch.manual_seed(0)
    for (cin, cout, k, stride, pad, hw) in [(2, 24, 4, 1, 0, (17, 13)), (24, 24, 3, 1, 0, (7, 5)), (24, 24, 2, 1, 0, (5, 3)), (3, 8, 3, 2, 1, (9, 8))]:
        ref = nn.Conv2d(cin, cout, k, stride=stride, padding=pad).to(device)
        mine = Conv2dAsGemm(cin, cout, k, stride=stride, padding=pad).to(device)
        mine.load_state_dict(ref.state_dict())
        assert list(mine.state_dict()) == list(ref.state_dict())
        x = torch.randn(n, cin, *hw, device=device)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = ref(xa), mine(xb)
        assert ya.shape == yb.shape
        torch.testing.assert_close(yb, ya, rtol=3e-5, atol=3e-5)
        g = torch.randn_like(ya)
        ya.backward(g), yb.backward(g)
        torch.testing.assert_close(xb.grad, xa.grad, rtol=3e-4, atol=3e-5)
        torch.testing.assert_close(mine.weight.grad, ref.weight.grad, rtol=3e-4, atol=3e-3)
        torch.testing.assert_close(mine.bias.grad, ref.bias.grad, rtol=3e-4, atol=3e-3)


def test_conv_as_gemm_formula_cpu(monkeypatch):
    # on the CPU the class defers to nn.Conv2d; push the GEMM formula through to check it there as well
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    _check("cpu", 1100)


def test_cnn_head_keeps_the_reference_parameter_names():
    m = CNN2dHead((2, 17, 13), (24, 24, 24), (4, 3, 2), (2, 1, 1), None, None, 64, "relu", True, None)
    assert [k for k in m.state_dict()][:4] == ["conv.conv.0.weight", "conv.conv.0.bias", "conv.conv.3.weight", "conv.conv.3.bias"]


@pytest.mark.gpu
def test_conv_as_gemm_matches_miopen_on_gpu():
    _check("cuda:0", 4096)
