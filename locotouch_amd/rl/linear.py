"""`nn.Linear` whose weight gradient is a split-K batched GEMM.

The PPO update runs every layer on minibatches of 24 576 rows (4096 envs x 24 steps / 4).  The weight gradient
dW[N, K] = dY[M, N]^T @ X[M, K] then has a tiny output (<= 512 x 348) and a 24 576-long reduction; issued as ONE GEMM,
hipBLASLt's stream-K kernels reach 20-50 TFLOP/s fp32 on MI355X (322 us for 512 x 256, 171 us for 348 x 512 - the two largest
kernels of the update, profiles/r02_kernel_stats.csv).  Cut into S = 8 row blocks and issued as a batched GEMM
[S][N, M/S] @ [S][M/S, K] followed by a sum over S, the same arithmetic takes 60 / 93 us (tools/wgrad_probe.py).
Forward and input gradient are the stock GEMMs (94-112 TFLOP/s).  Same parameters, same state_dict keys as `nn.Linear`;
the summation order of dW differs (relative difference ~3e-6, tests/test_rl_linear.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _SplitKLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, splits: int):
        ctx.save_for_backward(x, weight)
        ctx.splits = splits
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        s = ctx.splits
        m, n, k = x.shape[0], weight.shape[0], weight.shape[1]
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dyc = dy if dy.is_contiguous() else dy.contiguous()
            dw = torch.bmm(dyc.view(s, m // s, n).transpose(1, 2), x.view(s, m // s, k)).sum(0)
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class _SplitKMatmul(torch.autograd.Function):
    """`x @ w` for x [M, K], w [K, N] with the weight gradient x^T dy issued as a split-K batched GEMM (long reduction over M)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dy @ w.t() if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            m = x.shape[0]
            s = pick_splits(m)
            dyc = dy if dy.is_contiguous() else dy.contiguous()
            dw = (torch.bmm(x.view(s, m // s, -1).transpose(1, 2), dyc.view(s, m // s, -1)).sum(0) if s > 1 else x.t() @ dyc)
        return dx, dw


def split_k_matmul(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return _SplitKMatmul.apply(x, w)


def pick_splits(m: int) -> int:
    """Row blocks for the split-K weight gradient: the largest of a few small factors of m that leaves >= 64 rows per block."""
    for s in (8, 4, 5, 2, 3, 7):
        if m % s == 0 and m // s >= 64:
            return s
    return 1


class Linear(nn.Linear):
    """Drop-in `nn.Linear`; 2-D CUDA inputs with at least `split_k_min_rows` rows take the split-K weight-gradient path.

    `force_split_k`: take that path for every row count.  Set while a training step is being captured into a hipGraph: the
    stock weight-gradient GEMM (`dY^T @ X` as one hipBLASLt stream-K kernel) returns wrong results when REPLAYED from a graph
    at some shapes on this stack (3072 x 348 x 512: tools/ppo_graph_probe.py, SPLITK_MIN=1000000), the batched form does not.
    """

    split_k_min_rows = 4096
    force_split_k = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (x.dim() == 2 and x.is_cuda and (Linear.force_split_k or x.shape[0] >= self.split_k_min_rows) and x.is_contiguous()
                and torch.is_grad_enabled() and self.weight.requires_grad):
            return _SplitKLinear.apply(x, self.weight, self.bias, pick_splits(x.shape[0]))
        return F.linear(x, self.weight, self.bias)
