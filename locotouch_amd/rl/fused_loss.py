"""PPO minibatch loss through csrc/lt_ppo.hip: forward and gradients in ONE launch instead of ~105 small PyTorch launches.

`fused_ppo_loss(mu, std, value, batch..., cfg...)` returns (loss, surrogate, value_loss, entropy, kl) with autograd edges to
`mu`, `std` and `value`; the formulas are the reference's (loco_rl/loco_rl/algorithms/ppo.py:251-311), checked against the
PyTorch-op chain of `PPO._eager_update` in tests/test_hip_ppo_graph.py.  CUDA tensors, f32, state-independent ("scalar") std.
"""
from __future__ import annotations

import ctypes
import math

import torch

from .. import _abi


class _FusedPPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped):
        lib = _abi.load()
        m, a = mu.shape
        c = lambda t: t.detach().contiguous()  # noqa: E731
        mu_c, std_c, v_c = c(mu), c(std), c(value).view(-1)
        dmu = torch.empty_like(mu_c)
        dvalue = torch.empty(m, device=mu.device, dtype=torch.float32)
        acc = torch.empty(20, device=mu.device, dtype=torch.float32)
        vp = ctypes.c_void_p
        args = [c(actions), c(old_logp).view(-1), c(adv).view(-1), c(returns).view(-1), c(old_values).view(-1), c(old_mu), c(old_sigma)]
        _abi.check(lib.lt_ppo_loss(vp(mu_c.data_ptr()), vp(std_c.data_ptr()), vp(v_c.data_ptr()), *[vp(t.data_ptr()) for t in args], m, a,
                                   float(clip), float(vcoef), int(bool(clipped)), vp(dmu.data_ptr()), vp(dvalue.data_ptr()), vp(acc.data_ptr()),
                                   vp(torch.cuda.current_stream(mu.device).cuda_stream)), "lt_ppo_loss")
        surr, vl, kl = acc[0] / m, acc[1] / m, acc[2] / m
        ent = (0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(std_c)).sum()  # Normal entropy, summed over actions; same in every row
        loss = surr + vcoef * vl - ecoef * ent
        ctx.save_for_backward(dmu, dvalue.view_as(value), acc[4:4 + a] - ecoef / std_c)
        return loss, surr, vl, ent, kl

    @staticmethod
    def backward(ctx, g, *unused):
        dmu, dvalue, dstd = ctx.saved_tensors
        return g * dmu, g * dstd, g * dvalue, None, None, None, None, None, None, None, None, None, None, None


def fused_ppo_loss(mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip_param, value_loss_coef,
                   entropy_coef, use_clipped_value_loss):
    return _FusedPPOLoss.apply(mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip_param, value_loss_coef,
                               entropy_coef, use_clipped_value_loss)
