"""Single-layer GRU over a padded batch of whole trajectories, laid out for the distillation's shapes.

The BC epochs run the student's GRU (input 64, hidden 512) over L ~ 500 steps with only B ~ 47 trajectories per batch
(`batch_steps` 20 000 / episode length, reference distillation_cfg.py:60).  MIOpen's RNN path takes ~0.7 ms per time step
forward + backward at that shape (365 ms per batch - 95 % of the distillation's wall clock, profiles/r02_distill_demo_405.jsonl).
The recurrence itself is one [B, H] x [H, 3H] GEMM and one pointwise gate kernel per step; everything else is NOT sequential
and is hoisted out of the loop here:

  forward    igates = X W_ih^T for ALL steps in one GEMM; per step: hgates = h W_hh^T (GEMM), fused gate kernel
  backward   per step: fused gate-backward kernel, dh += dgates_h W_hh (GEMM); after the loop ONE GEMM each for dW_hh
             (sum over all L*B rows, split-K), dW_ih, dX and the bias sums

The gate arithmetic is PyTorch's own fused cell kernel (`_thnn_fused_gru_cell`, what `torch.gru_cell` runs) on CUDA/HIP and the
same formulas in plain ops on the CPU; results equal `nn.GRU` to fp32 rounding (tests/test_rl_gru.py).  Parameters are the
`nn.GRU` module's own (`weight_ih_l0`, ...): checkpoints are unaffected.
"""
from __future__ import annotations

import torch


def _cell_forward(ig, hg, h, b_ih, b_hh):
    """(h', workspace) of one step from the bias-free gate pre-activations."""
    if ig.is_cuda:
        return torch.ops.aten._thnn_fused_gru_cell(ig, hg, h, b_ih, b_hh)
    hgb = hg + b_hh
    i_r, i_z, i_n = (ig + b_ih).chunk(3, 1)
    h_r, h_z, h_n = hgb.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return n + z * (h - n), torch.cat([r, z, n, h, h_n], dim=1)  # workspace layout of the fused kernel: r, z, n, hx, h_n (+bias)


def _cell_backward(dh, ws, hidden):
    """(d igates, d hgates, d h_prev) from d h' and the step's workspace."""
    if dh.is_cuda:
        dig, dhg, dhx, _, _ = torch.ops.aten._thnn_fused_gru_cell_backward(dh, ws, True)
        return dig, dhg, dhx
    r, z, n, hx, h_n = ws.split(hidden, dim=1)
    dn = dh * (1 - z)
    dz = dh * (hx - n)
    dn_pre = dn * (1 - n * n)
    dz_pre = dz * z * (1 - z)
    dr_pre = dn_pre * h_n * r * (1 - r)
    return torch.cat([dr_pre, dz_pre, dn_pre], 1), torch.cat([dr_pre, dz_pre, dn_pre * r], 1), dh * z


def _wgrad(dy2d, x2d):
    """dy^T @ x with a long reduction: split-K as a batched GEMM + sum (see rl/linear.py; also what is safe under graph replay)."""
    from .linear import pick_splits

    m = dy2d.shape[0]
    s = pick_splits(m) if dy2d.is_cuda else 1
    if s > 1:
        return torch.bmm(dy2d.view(s, m // s, -1).transpose(1, 2), x2d.view(s, m // s, -1)).sum(0)
    return dy2d.t() @ x2d


class _GRUSequence(torch.autograd.Function):
    """Time loop in PyTorch ops (CPU, and the GPU when the HIP kernels do not apply)."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh):
        L, B, _ = x.shape
        H = w_hh.shape[1]
        ig = (x.reshape(L * B, -1) @ w_ih.t()).view(L, B, 3 * H)
        w_hh_t = w_hh.t().contiguous()
        out = x.new_empty(L, B, H)
        ws = x.new_empty(L, B, 5 * H)
        h = h0
        for t in range(L):
            h, w = _cell_forward(ig[t], h @ w_hh_t, h, b_ih, b_hh)
            out[t] = h
            ws[t] = w
        ctx.save_for_backward(x, h0, w_ih, w_hh, out, ws)
        return out, h

    @staticmethod
    def backward(ctx, dout, dhn):
        x, h0, w_ih, w_hh, out, ws = ctx.saved_tensors
        L, B, _ = x.shape
        H = w_hh.shape[1]
        dig = x.new_empty(L, B, 3 * H)
        dhg = x.new_empty(L, B, 3 * H)
        dh = dhn.clone() if dhn is not None else x.new_zeros(B, H)
        for t in range(L - 1, -1, -1):
            a, b, dhx = _cell_backward(dh + dout[t], ws[t], H)
            dig[t] = a
            dhg[t] = b
            dh = torch.addmm(dhx, b, w_hh)
        return _finish_backward(ctx, x, h0, w_ih, out, dig, dhg, dh)


def _finish_backward(ctx, x, h0, w_ih, out, dig, dhg, dh0):
    """Everything of the backward pass that is not sequential: one GEMM each."""
    L, B, _ = x.shape
    H = out.shape[2]
    dig2, dhg2 = dig.view(L * B, 3 * H), dhg.view(L * B, 3 * H)
    h_prev = torch.cat([h0.unsqueeze(0), out[:-1]], dim=0).view(L * B, H)
    dw_hh = _wgrad(dhg2, h_prev)
    dw_ih = _wgrad(dig2, x.reshape(L * B, -1))
    dx = (dig2 @ w_ih).view_as(x) if ctx.needs_input_grad[0] else None
    return dx, dh0, dw_ih, dw_hh, dig2.sum(0), dhg2.sum(0)


class _GRUSequenceHip(torch.autograd.Function):
    """Time loop in csrc/lt_gru.hip: one launch per step forward, two backward, issued from C (`lt_gru_forward/_backward`)."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh):
        import ctypes

        from .. import _abi

        lib = _abi.load()
        L, B, _ = x.shape
        H = w_hh.shape[1]
        ig = (x.reshape(L * B, -1) @ w_ih.t()).view(L, B, 3 * H)
        h0c, w_hh_c = h0.contiguous(), w_hh.contiguous()
        out = x.new_empty(L, B, H)
        ws = x.new_empty(L, B, 4 * H)
        vp = ctypes.c_void_p
        stream = vp(torch.cuda.current_stream(x.device).cuda_stream)
        _abi.check(lib.lt_gru_forward(vp(ig.data_ptr()), vp(h0c.data_ptr()), vp(w_hh_c.data_ptr()), vp(b_ih.data_ptr()), vp(b_hh.data_ptr()),
                                      L, B, H, vp(out.data_ptr()), vp(ws.data_ptr()), stream), "lt_gru_forward")
        ctx.save_for_backward(x, h0c, w_ih, w_hh_c, out, ws)
        return out, out[-1]

    @staticmethod
    def backward(ctx, dout, dhn):
        import ctypes

        from .. import _abi

        lib = _abi.load()
        x, h0, w_ih, w_hh, out, ws = ctx.saved_tensors
        L, B, _ = x.shape
        H = w_hh.shape[1]
        dout = dout.contiguous()
        dig = x.new_empty(L, B, 3 * H)
        dhg = x.new_empty(L, B, 3 * H)
        scratch = x.new_empty(3, B, H)
        dh0 = x.new_empty(B, H)
        vp = ctypes.c_void_p
        stream = vp(torch.cuda.current_stream(x.device).cuda_stream)
        dhn_p = vp(dhn.contiguous().data_ptr()) if dhn is not None else vp(None)
        _abi.check(lib.lt_gru_backward(vp(dout.data_ptr()), dhn_p, vp(out.data_ptr()), vp(ws.data_ptr()), vp(h0.data_ptr()), vp(w_hh.data_ptr()),
                                       L, B, H, vp(dig.data_ptr()), vp(dhg.data_ptr()), vp(scratch.data_ptr()), vp(dh0.data_ptr()), stream),
                   "lt_gru_backward")
        return _finish_backward(ctx, x, h0, w_ih, out, dig, dhg, dh0)


def gru_sequence(gru: torch.nn.GRU, x: torch.Tensor, h0: torch.Tensor | None = None):
    """`gru(x, h0)` for a single-layer, unidirectional, time-major `nn.GRU`: (output [L, B, H], h_n [1, B, H])."""
    assert gru.num_layers == 1 and not gru.bidirectional and not gru.batch_first and gru.bias
    h = h0[0] if h0 is not None else x.new_zeros(x.shape[1], gru.hidden_size)
    fn = _GRUSequenceHip if (x.is_cuda and x.dtype == torch.float32 and gru.hidden_size % 64 == 0 and use_hip_kernels) else _GRUSequence
    out, hn = fn.apply(x.contiguous(), h, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    return out, hn.unsqueeze(0)


use_hip_kernels = True  # False: the PyTorch-op time loop on the GPU as well (tools/gru_probe.py compares the three forms)
