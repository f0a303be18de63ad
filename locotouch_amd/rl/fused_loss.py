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
    def forward(ctx, mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped, idx):
        lib = _abi.load()
        m, a = mu.shape
        c = lambda t: t.detach().contiguous()  # noqa: E731
        mu_c, std_c, v_c = c(mu), c(std), c(value).view(-1)
        dmu = torch.empty_like(mu_c)
        dvalue = torch.empty(m, device=mu.device, dtype=torch.float32)
        acc = torch.empty(24, device=mu.device, dtype=torch.float32)
        vp = ctypes.c_void_p
        args = [c(actions), c(old_logp).view(-1), c(adv).view(-1), c(returns).view(-1), c(old_values).view(-1), c(old_mu), c(old_sigma)]
        rows = args[0].shape[0]
        if (any(t.shape[0] != rows for t in args) or any(t.shape != (rows, a) for t in (args[0], args[5], args[6]))
                or (idx is None and rows != m) or (idx is not None and (idx.dtype != torch.int64 or idx.numel() != m))):
            raise ValueError("fused_ppo_loss: batch tensors do not match the minibatch / index")
        idx_c = None if idx is None else c(idx)
        out = torch.empty(24, device=mu.device, dtype=torch.float32)
        _abi.check(lib.lt_ppo_loss(vp(mu_c.data_ptr()), vp(std_c.data_ptr()), vp(v_c.data_ptr()), *[vp(t.data_ptr()) for t in args],
                                   vp(None if idx_c is None else idx_c.data_ptr()), m, a,
                                   float(clip), float(vcoef), float(ecoef), int(bool(clipped)), vp(dmu.data_ptr()), vp(dvalue.data_ptr()), vp(acc.data_ptr()),
                                   vp(out.data_ptr()), vp(torch.cuda.current_stream(mu.device).cuda_stream)), "lt_ppo_loss")
        # out: the finished scalars, written by a one-wave launch behind the main kernel (a dozen 12-float tensor ops otherwise)
        ctx.save_for_backward(dmu, dvalue.view_as(value), out[8:8 + a])
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, g, *unused):
        dmu, dvalue, dstd = ctx.saved_tensors
        return g * dmu, g * dstd, g * dvalue, None, None, None, None, None, None, None, None, None, None, None, None


def fused_ppo_loss(mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip_param, value_loss_coef,
                   entropy_coef, use_clipped_value_loss, idx=None):
    """`idx` (int64 [M], optional): the batch tensors `actions ... old_sigma` are then the WHOLE flattened rollout storage and
    minibatch row i is their row idx[i] - the kernel gathers while it loads, seven gather launches less per step."""
    return _FusedPPOLoss.apply(mu, std, value, actions, old_logp, adv, returns, old_values, old_mu, old_sigma, clip_param, value_loss_coef,
                               entropy_coef, use_clipped_value_loss, idx)
