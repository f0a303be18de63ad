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
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        if ctx.needs_input_grad[1] and _head_wgrad_ok(dyc, x):
            dw, db = _head_wgrad(dyc, x)  # narrow heads (12 / 1 outputs): weight and bias gradient in one pass over x
            return dx, dw, (db if ctx.has_bias and ctx.needs_input_grad[2] else None), None
        dw = _wgrad(dyc, x, s) if ctx.needs_input_grad[1] else None
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


def pick_splits(m: int, n: int | None = None, k: int | None = None) -> int:
    """Row blocks for the split-K weight gradient dW[n, k] = dY[m, n]^T X[m, k].

    Without the output shape: the largest of a few small factors of m that leaves >= 64 rows per block.  With it: enough blocks
    that (blocks x 64 x 64 output tiles) cover the chip's 256 CUs, between 8 and 16 (tools/wgrad_probe.py), with at least 128
    rows per block.  (Narrow heads, n <= 16, do not come here on the GPU: `_head_wgrad`.)"""
    if n is not None and k is not None:
        tiles = -(-n // 64) * -(-k // 64)
        want = max(8, min(16, -(-256 // tiles)))  # (measured: 128 x 256 takes 27 us at 8-16 blocks, 94 us at 32; 512 x 348 is flat 4-16)
        best = 1
        for s in (2, 3, 4, 6, 8, 12, 16):
            if m % s == 0 and m // s >= 128 and s <= want:
                best = s
        if best > 1 or m < 256:
            return best
    for s in (8, 4, 5, 2, 3, 7):
        if m % s == 0 and m // s >= 64:
            return s
    return 1


def _wgrad(dy: torch.Tensor, x: torch.Tensor, s: int) -> torch.Tensor:
    m, n, k = x.shape[0], dy.shape[1], x.shape[1]
    if s <= 1:
        return dy.t() @ x
    return torch.bmm(dy.view(s, m // s, n).transpose(1, 2), x.view(s, m // s, k)).sum(0)


def _head_wgrad_ok(dy: torch.Tensor, x: torch.Tensor) -> bool:
    n, k = dy.shape[1], x.shape[1]
    if not (dy.is_cuda and dy.dtype == torch.float32 and x.dtype == torch.float32 and n <= 16 and k % 4 == 0 and 4 <= k <= 1024 and x.is_contiguous()):
        return False
    lanes, nn = 256 // (k // 4), (1 if n <= 1 else 4 if n <= 4 else 8 if n <= 8 else 12 if n <= 12 else 16)
    return lanes * nn * (k // 4) * 16 + lanes * nn * 4 <= 72 * 1024  # one block's LDS partials (csrc/lt_ppo.hip)


def _head_wgrad(dy: torch.Tensor, x: torch.Tensor):
    import ctypes

    from .. import _abi

    lib = _abi.load()
    m, n, k = x.shape[0], dy.shape[1], x.shape[1]
    dw = torch.empty(n, k, device=x.device, dtype=torch.float32)
    db = torch.empty(n, device=x.device, dtype=torch.float32)
    ws = torch.empty(int(lib.lt_head_wgrad_ws_floats(m, n, k)), device=x.device, dtype=torch.float32)
    vp = ctypes.c_void_p
    _abi.check(lib.lt_head_wgrad(vp(dy.data_ptr()), vp(x.data_ptr()), 0, m, n, k, vp(dw.data_ptr()), vp(db.data_ptr()), vp(ws.data_ptr()),
                                 vp(torch.cuda.current_stream(x.device).cuda_stream)), "lt_head_wgrad")
    return dw, db


class _LinearELU(torch.autograd.Function):
    """`elu(linear(x))` as one autograd node on the GPU: backward recovers elu' from the saved OUTPUT and gets the bias gradient
    from the same pass (csrc/lt_ppo.hip `lt_elu_backward_bias`: PyTorch runs an elementwise kernel and a 24 576-row column
    reduction per layer for these), then the split-K weight gradient and the input gradient as GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha: float, splits: int):
        a = F.elu(F.linear(x, weight, bias), alpha, inplace=True)
        ctx.save_for_backward(x, weight, a)
        ctx.alpha, ctx.splits = alpha, splits
        return a

    @staticmethod
    def backward(ctx, da):
        import ctypes

        from .. import _abi

        x, weight, a = ctx.saved_tensors
        lib = _abi.load()
        m, n = a.shape
        da = da if da.is_contiguous() else da.contiguous()
        dz = torch.empty_like(a)
        db = torch.empty(n, device=a.device, dtype=a.dtype)
        ws = torch.empty(int(lib.lt_elu_backward_bias_ws_floats(m, n)), device=a.device, dtype=torch.float32)
        vp = ctypes.c_void_p
        _abi.check(lib.lt_elu_backward_bias(vp(da.data_ptr()), vp(a.data_ptr()), m, n, float(ctx.alpha), vp(dz.data_ptr()), vp(db.data_ptr()),
                                            vp(ws.data_ptr()), vp(torch.cuda.current_stream(a.device).cuda_stream)), "lt_elu_backward_bias")
        dx = dz @ weight if ctx.needs_input_grad[0] else None
        dw = _wgrad(dz, x, ctx.splits) if ctx.needs_input_grad[1] else None
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None, None


def linear_elu_ok(x: torch.Tensor, lin: nn.Linear, alpha_module) -> bool:
    n = lin.out_features
    return (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and torch.is_grad_enabled() and lin.weight.requires_grad
            and lin.bias is not None and x.shape[0] >= Linear.split_k_min_rows and n % 4 == 0 and n <= 1024 and not alpha_module.inplace)


def linear_elu(x: torch.Tensor, lin: nn.Linear, alpha: float) -> torch.Tensor:
    return _LinearELU.apply(x, lin.weight, lin.bias, float(alpha), pick_splits(x.shape[0], lin.out_features, lin.in_features))


class MLPSequential(nn.Sequential):
    """`nn.Sequential` (same children, same state_dict keys) whose forward runs `Linear -> ELU` pairs of large training batches
    as the fused node above; everything else (inference, small batches, other activations, CPU) is the plain module walk."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            mod = mods[i]
            if (i + 1 < len(mods) and isinstance(mod, nn.Linear) and type(mods[i + 1]) is nn.ELU and linear_elu_ok(x, mod, mods[i + 1])):
                x = linear_elu(x, mod, mods[i + 1].alpha)
                i += 2
            else:
                x = mod(x)
                i += 1
        return x


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
            return _SplitKLinear.apply(x, self.weight, self.bias, pick_splits(x.shape[0], self.out_features, self.in_features))
        return F.linear(x, self.weight, self.bias)
