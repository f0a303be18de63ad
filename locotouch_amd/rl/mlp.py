"""Single-launch MLP inference on the HIP library (locotouch_amd/csrc/lt_mlp.hip) for an `nn.Sequential` of Linear layers
with one activation between them - the actor / critic of the reference's ActorCritic
(loco_rl/loco_rl/modules/actor_critic.py:41-67 builds exactly such stacks).  Arithmetic: f32-equivalent - both operands
are split into fp16 (hi, lo) pairs and multiplied on the fp16 MFMA with an f32 error-compensation term (three MFMAs per
tile, DESIGN.md "MLP kernel"); parameters and results are f32, the error is below an f32 GEMM's own rounding.

The packed parameter buffer is refreshed from the live `nn.Linear` parameters by `pack()` (a few small launches; call it
whenever the optimizer has stepped, e.g. at the start of every rollout - it is stream-ordered and graph-capturable).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _abi

_ACT_IDS = {nn.ELU: "LT_ACT_ELU", nn.ReLU: "LT_ACT_RELU", nn.Tanh: "LT_ACT_TANH", nn.Identity: "LT_ACT_NONE"}


def describe(seq: nn.Sequential):
    """(LtMlpDesc, [Linear...]) for a supported stack, or None (caller keeps the torch path)."""
    linears, acts = [], set()
    mods = list(seq)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            if m.bias is None or m.weight.dtype != torch.float32:
                return None
            linears.append(m)
        elif type(m) in _ACT_IDS:
            if type(m) is nn.ELU and (m.alpha != 1.0):
                return None
            if i == 0 or not isinstance(mods[i - 1], nn.Linear) or i == len(mods) - 1:
                return None
            acts.add(type(m))
        else:
            return None
    C = _abi.CONSTS
    if not linears or len(linears) > C["LT_MLP_MAX_LAYERS"] or len(acts) > 1:
        return None
    if len(linears) > 1 and len(mods) != 2 * len(linears) - 1:
        return None  # an activation must follow every hidden layer
    dims = [linears[0].in_features] + [l.out_features for l in linears]
    if dims[0] > C["LT_MLP_MAX_WIDTH"] or any(d > 512 for d in dims[1:]):
        return None
    if any(a.out_features != b.in_features for a, b in zip(linears[:-1], linears[1:])):
        return None
    desc = _abi.LtMlpDesc()
    desc.num_layers = len(linears)
    for i, d in enumerate(dims):
        desc.dims[i] = d
    desc.activation = C[_ACT_IDS[next(iter(acts))]] if acts else C["LT_ACT_NONE"]
    return desc, linears


class PackedMLP:
    def __init__(self, seq: nn.Sequential):
        got = describe(seq)
        if got is None:
            raise ValueError("network shape not supported by lt_mlp_forward (Linear/activation stack, hidden widths <= 512)")
        self.desc, self.linears = got
        self.lib = _abi.load()
        n = ctypes.c_size_t()
        _abi.check(self.lib.lt_mlp_packed_floats(ctypes.byref(self.desc), ctypes.byref(n)), "lt_mlp_packed_floats")
        dev = self.linears[0].weight.device
        self.packed = torch.zeros(int(n.value), device=dev, dtype=torch.float32)
        self.out_features = self.linears[-1].out_features
        self.in_features = self.linears[0].in_features
        self.pack()

    @staticmethod
    def _stream() -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def pack(self) -> None:
        L = len(self.linears)
        arr = ctypes.c_void_p * L
        w = arr(*[l.weight.data_ptr() for l in self.linears])
        b = arr(*[l.bias.data_ptr() for l in self.linears])
        for l in self.linears:
            if not l.weight.is_contiguous():
                raise ValueError("Linear.weight must be contiguous")
        _abi.check(self.lib.lt_mlp_pack(ctypes.byref(self.desc), w, b, ctypes.c_void_p(self.packed.data_ptr()), self._stream()), "lt_mlp_pack")

    def backward_ok(self) -> bool:
        """The chain of input gradients can run through the MLP kernel (lt_mlp_backward_pair): ELU, hidden widths % 8, <= 64 outputs."""
        n = ctypes.c_size_t()
        return self.lib.lt_mlp_backward_packed_floats(ctypes.byref(self.desc), ctypes.byref(n)) == 0

    def pack_backward(self) -> None:
        """The transposed weights in the kernel's stream layout (the backward chain multiplies by W^T); re-packed at every optimizer step."""
        if getattr(self, "bpacked", None) is None:
            n = ctypes.c_size_t()
            _abi.check(self.lib.lt_mlp_backward_packed_floats(ctypes.byref(self.desc), ctypes.byref(n)), "lt_mlp_backward_packed_floats")
            self.bpacked = torch.zeros(int(n.value), device=self.packed.device, dtype=torch.float32)
        L = len(self.linears)
        w = (ctypes.c_void_p * L)(*[l.weight.data_ptr() for l in self.linears])
        _abi.check(self.lib.lt_mlp_pack_backward(ctypes.byref(self.desc), w, ctypes.c_void_p(self.bpacked.data_ptr()), self._stream()), "lt_mlp_pack_backward")

    def set_input_format(self, dtype: torch.dtype) -> None:
        """float32 rows (default) or bfloat16 rows (BASELINE config 5: widened exactly to f32 inside the kernel)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("input rows are float32 or bfloat16")
        if dtype == torch.bfloat16 and self.in_features % 4:
            raise ValueError("bfloat16 input rows need in_features % 4 == 0")
        self.desc.input_format = _abi.CONSTS["LT_ROWS_BF16"] if dtype == torch.bfloat16 else _abi.CONSTS["LT_ROWS_F32"]
        self.in_dtype = dtype

    def forward(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if x.dim() != 2 or x.shape[1] != self.in_features or x.dtype != getattr(self, "in_dtype", torch.float32) or not x.is_contiguous():
            raise ValueError("x must be a contiguous [m, in_features] tensor of the input format")
        m = x.shape[0]
        if out is None:
            out = torch.empty(m, self.out_features, device=x.device, dtype=torch.float32)
        _abi.check(self.lib.lt_mlp_forward(ctypes.byref(self.desc), ctypes.c_void_p(self.packed.data_ptr()), ctypes.c_void_p(x.data_ptr()), m,
                                           ctypes.c_void_p(out.data_ptr()), self._stream()), "lt_mlp_forward")
        return out

    __call__ = forward


class _PairForward(torch.autograd.Function):
    """Actor and critic forward of a PPO minibatch in ONE launch of the MFMA MLP kernel (csrc/lt_mlp.hip `lt_mlp_forward_pair`),
    with the hidden activations written out for the backward pass; backward is the layer chain by hand: ELU' + bias sums
    (`lt_elu_backward_bias`), narrow-head gradients (`lt_head_wgrad`), split-K weight gradients and the input-gradient GEMMs
    (rl/linear.py).  Inputs: x0, x1, then the (weight, bias) pairs of both stacks, so that autograd routes the gradients."""

    @staticmethod
    def forward(ctx, pair, x0, x1, *params):
        lib = _abi.load()
        vp = ctypes.c_void_p
        m = x0.shape[0]
        nets = (pair.a, pair.b)
        ys, acts = [], []
        for net in nets:
            dims = [net.desc.dims[i] for i in range(net.desc.num_layers + 1)]
            ys.append(torch.empty(m, dims[-1], device=x0.device, dtype=torch.float32))
            acts.append([torch.empty(m, d, device=x0.device, dtype=torch.float32) for d in dims[1:-1]])
        arr = [(vp * max(1, len(a)))(*[t.data_ptr() for t in a]) for a in acts]
        _abi.check(lib.lt_mlp_forward_pair(ctypes.byref(nets[0].desc), vp(nets[0].packed.data_ptr()), vp(x0.data_ptr()),
                                           ctypes.byref(nets[1].desc), vp(nets[1].packed.data_ptr()), vp(x1.data_ptr()), m,
                                           vp(ys[0].data_ptr()), vp(ys[1].data_ptr()), arr[0], arr[1], 0, PackedMLP._stream()), "lt_mlp_forward_pair")
        if pair.check_domain:  # once per PPO update (PackedPair.arm_domain_check): largest |value| that entered any layer
            pair.check_domain = False
            with torch.no_grad():
                m_ = torch.stack([t.abs().max() for t in (x0, x1, *acts[0], *acts[1])]).max()
                pair.domain_max = m_ if pair.domain_max is None else torch.maximum(pair.domain_max, m_)
        ctx.nl = (len(nets[0].linears), len(nets[1].linears))
        ctx.save_for_backward(x0, x1, *acts[0], *acts[1], *[p for p in params[0::2]])
        ctx.alpha = 1.0
        ctx.elu = nets[0].desc.activation == _abi.CONSTS["LT_ACT_ELU"]
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        from .linear import _head_wgrad, _head_wgrad_ok, _wgrad, pick_splits

        lib = _abi.load()
        vp = ctypes.c_void_p
        saved = list(ctx.saved_tensors)
        x = saved[:2]
        n0, n1 = ctx.nl
        acts = [saved[2:2 + n0 - 1], saved[2 + n0 - 1:2 + n0 - 1 + n1 - 1]]
        ws = saved[2 + n0 - 1 + n1 - 1:]
        weights = [ws[:n0], ws[n0:]]
        grads = []
        for k, dy in enumerate((dy0, dy1)):
            L = len(weights[k])
            g = dy if dy.is_contiguous() else dy.contiguous()
            layer_grads = [None] * (2 * L)
            for l in range(L - 1, -1, -1):
                inp = acts[k][l - 1] if l > 0 else x[k]
                w = weights[k][l]
                m, n = g.shape
                if l < L - 1:  # g is the gradient w.r.t. the activation output: through ELU', with the bias sums in the same pass
                    a = acts[k][l]
                    dz = torch.empty_like(a)
                    db = torch.empty(n, device=a.device, dtype=torch.float32)
                    scratch = torch.empty(int(lib.lt_elu_backward_bias_ws_floats(m, n)), device=a.device, dtype=torch.float32)
                    _abi.check(lib.lt_elu_backward_bias(vp(g.data_ptr()), vp(a.data_ptr()), m, n, 1.0, vp(dz.data_ptr()), vp(db.data_ptr()), vp(scratch.data_ptr()),
                                                        PackedMLP._stream()), "lt_elu_backward_bias")
                    dw = _wgrad(dz, inp, pick_splits(m, n, inp.shape[1]))
                else:  # the head: no activation
                    dz = g
                    if _head_wgrad_ok(dz, inp):
                        dw, db = _head_wgrad(dz, inp)
                    else:
                        dw, db = _wgrad(dz, inp, pick_splits(m, n, inp.shape[1])), dz.sum(0)
                layer_grads[2 * l], layer_grads[2 * l + 1] = dw, db
                if l > 0:
                    g = dz @ w
            grads += layer_grads
        return (None, None, None, *grads)


def _alloc_outputs(nets, m, device):
    ys, acts = [], []
    for net in nets:
        dims = [net.desc.dims[i] for i in range(net.desc.num_layers + 1)]
        ys.append(torch.empty(m, dims[-1], device=device, dtype=torch.float32))
        acts.append([torch.empty(m, d, device=device, dtype=torch.float32) for d in dims[1:-1]])
    return ys, acts


import os as _os

USE_SPLIT_F16_WGRAD = _os.environ.get("LT_SPLIT_F16_WGRAD", "1") != "0"  # 0: the weight gradients as library f32 GEMMs (A/B measurements)
USE_SPLIT_DZ = _os.environ.get("LT_SPLIT_DZ", "1") != "0"  # 0: the chain writes f32 dz (one scale per workgroup), lt_wgrad splits it on the fly
USE_FUSED_BACKWARD = _os.environ.get("LT_FUSED_BACKWARD", "1") != "0"  # 0: dz @ W as library GEMMs + lt_elu_backward_bias per layer


class SumJobs:
    """Ordered partial sums collected over a backward pass and added by ONE launch (`lt_partial_sums`): the per-block bias sums of
    lt_elu_backward_bias / lt_head_wgrad and the split-K slabs of the weight-gradient GEMMs were 14 launches of ~5 us per
    minibatch step."""

    MAX = 24

    def __init__(self):
        self.jobs: list = []   # (ws tensor, nblk, stride, count, split, out0, out1 | None)
        self.after: list = []  # (dst, src) copies to run behind the launch (a padded sum's useful columns)

    def add(self, ws, nblk, stride, count, split, out0, out1=None) -> None:
        self.jobs.append((ws, int(nblk), int(stride), int(count), int(split), out0, out1))

    def launch(self) -> None:
        lib = _abi.load()
        vp = ctypes.c_void_p
        while self.jobs:
            batch, self.jobs = self.jobs[:self.MAX], self.jobs[self.MAX:]
            n = len(batch)
            arr_p = lambda vals: (vp * n)(*vals)  # noqa: E731
            ws = arr_p([j[0].data_ptr() for j in batch])
            out0 = arr_p([j[5].data_ptr() for j in batch])
            out1 = arr_p([None if j[6] is None else j[6].data_ptr() for j in batch])
            nblk = (ctypes.c_int * n)(*[j[1] for j in batch])
            stride = (ctypes.c_int64 * n)(*[j[2] for j in batch])
            count = (ctypes.c_int * n)(*[j[3] for j in batch])
            split = (ctypes.c_int * n)(*[j[4] for j in batch])
            _abi.check(lib.lt_partial_sums(n, ws, nblk, stride, count, split, out0, out1, PackedMLP._stream()), "lt_partial_sums")
            self._keep = batch  # the launch reads the buffers asynchronously
        for dst, src in self.after:
            dst.copy_(src)
        self.after = []


def backward_chain(weights, biases_out, weights_out, x, acts, dy, sums: SumJobs):
    """The backward pass of one Linear/ELU stack by hand, gradients written IN PLACE into the given tensors (views of the flat
    gradient bucket): `weights[l]` the layer's weight, `weights_out[l]` / `biases_out[l]` where dW_l / db_l go, `x` the stack's
    input rows, `acts[l]` the activations behind hidden layer l, `dy` the gradient w.r.t. the stack's output.  ELU' + per-block
    bias sums in one kernel (lt_elu_backward_bias), narrow-head gradients in one (lt_head_wgrad), split-K batched GEMMs for the
    weight gradients, plain GEMMs for the input gradients; every ordered sum of partials is queued on `sums` (one launch later)."""
    from .linear import _head_wgrad_ok, pick_splits

    lib = _abi.load()
    vp = ctypes.c_void_p
    stream = PackedMLP._stream()
    L = len(weights)
    g = dy if dy.is_contiguous() else dy.contiguous()
    for l in range(L - 1, -1, -1):
        inp = acts[l - 1] if l > 0 else x
        w = weights[l]
        m, n = g.shape
        k = inp.shape[1]
        if l < L - 1:  # g is the gradient w.r.t. the activation output: through ELU', with the bias sums in the same pass
            a = acts[l]
            dz = torch.empty_like(a)
            nblk = int(lib.lt_elu_backward_bias_nblk(m))
            scratch = torch.empty(nblk * n + nblk, device=a.device, dtype=torch.float32)
            amax = scratch[nblk * n:]  # per-block max |dz|: lt_wgrad scales the gradient into f16's range by it
            _abi.check(lib.lt_elu_backward_bias2(vp(g.data_ptr()), vp(a.data_ptr()), m, n, 1.0, vp(dz.data_ptr()), vp(None),
                                                 vp(scratch.data_ptr()), vp(amax.data_ptr()), stream), "lt_elu_backward_bias2")
            sums.add(scratch, nblk, n, n, n, biases_out[l])
        else:
            dz, amax = g, None
        if l == L - 1 and _head_wgrad_ok(dz, inp):
            nblk = int(lib.lt_head_wgrad_nblk(m))
            ws = torch.empty(int(lib.lt_head_wgrad_ws_floats(m, n, k)), device=inp.device, dtype=torch.float32)
            _abi.check(lib.lt_head_wgrad(vp(dz.data_ptr()), vp(inp.data_ptr()), 0, m, n, k, vp(None), vp(None), vp(ws.data_ptr()), stream), "lt_head_wgrad")
            sums.add(ws, nblk, n * k + 16, n * k + n, n * k, weights_out[l], biases_out[l])
        elif amax is not None and USE_SPLIT_F16_WGRAD and n % 4 == 0 and k % 4 == 0:
            # dW = dz^T x on the f16 matrix cores, f32-equivalent (csrc/lt_wgrad.hip): slices of the rows -> slabs -> the joint sum launch
            sp = int(lib.lt_wgrad_splits(m, n, k))
            slabs = torch.empty(sp * n * k, device=inp.device, dtype=torch.float32)
            _abi.check(lib.lt_wgrad(vp(dz.data_ptr()), 0, vp(None), vp(inp.data_ptr()), 0, m, n, k, vp(amax.data_ptr()), amax.numel(), vp(slabs.data_ptr()), vp(None), stream), "lt_wgrad")
            sums.add(slabs, sp, n * k, n * k, n * k, weights_out[l])
        else:
            sp = pick_splits(m, n, k)
            if sp > 1:
                slabs = torch.bmm(dz.view(sp, m // sp, n).transpose(1, 2), inp.view(sp, m // sp, k))
                sums.add(slabs, sp, n * k, n * k, n * k, weights_out[l])
            else:
                torch.mm(dz.t(), inp, out=weights_out[l])
            if l == L - 1:
                torch.sum(dz, dim=0, out=biases_out[l])
        if l > 0:
            g = dz @ w


class PackedPair:
    """The actor and critic stacks of an ActorCritic as one training forward (see `_PairForward`).  `__call__(obs, critic_obs)`
    re-packs the live parameters (they change at every optimizer step), runs the launch and returns (mean, value) with autograd
    edges to every Linear's weight and bias."""

    def __init__(self, actor: nn.Sequential, critic: nn.Sequential):
        self.a, self.b = PackedMLP(actor), PackedMLP(critic)
        # domain monitor of the MLP kernel (include/lt_env.h, LT_MLP_INPUT_CLAMP): layer inputs beyond +-1000 are saturated by the
        # kernel; the next forward after arm_domain_check() records the largest |input row / hidden activation| on the device
        self.check_domain = False
        self.domain_max: torch.Tensor | None = None
        for net in (self.a, self.b):
            dims = [net.desc.dims[i] for i in range(net.desc.num_layers + 1)]
            if any(d % 4 for d in dims[1:-1]) or net.desc.activation != _abi.CONSTS["LT_ACT_ELU"]:
                raise ValueError("PackedPair: ELU stacks with hidden widths that are multiples of 4")

    def arm_domain_check(self) -> None:
        self.check_domain = True

    def domain_violated(self) -> bool:
        """True if a checked forward saw a layer input at or beyond the kernel's saturation bound (host read: call where the
        trainer reads its statistics anyway).  Resets the record."""
        if self.domain_max is None:
            return False
        v = float(self.domain_max)
        self.domain_max = None
        return v >= float(_abi.CONSTS["LT_MLP_INPUT_CLAMP"])

    def forward_raw(self, x0: torch.Tensor, x1: torch.Tensor, split: bool | None = None):
        """The launch without autograd: ((mean, value), (activations of the actor, of the critic)); re-packs the live parameters.
        `split` (default: whenever the fused backward pass will consume them): the activations are written in the kernel's split
        format - one dword per element, f16 hi | f16 lo << 16, value = hi + lo / 64 (include/lt_env.h, lt_mlp_forward_pair) - which
        the backward chain and the weight-gradient kernel read without converting; the tensors keep dtype float32 as a container."""
        lib = _abi.load()
        vp = ctypes.c_void_p
        nets = (self.a, self.b)
        m = x0.shape[0]
        if split is None:
            split = self._fused_backward_possible(x0, x1)
        self.acts_split = bool(split)
        self.pack_training(with_backward=self.acts_split)
        ys, acts = _alloc_outputs(nets, m, x0.device)
        arr = [(vp * max(1, len(a)))(*[t.data_ptr() for t in a]) for a in acts]
        _abi.check(lib.lt_mlp_forward_pair(ctypes.byref(nets[0].desc), vp(nets[0].packed.data_ptr()), vp(x0.data_ptr()),
                                           ctypes.byref(nets[1].desc), vp(nets[1].packed.data_ptr()), vp(x1.data_ptr()), m,
                                           vp(ys[0].data_ptr()), vp(ys[1].data_ptr()), arr[0], arr[1], int(self.acts_split), PackedMLP._stream()), "lt_mlp_forward_pair")
        if self.check_domain:
            self.check_domain = False
            hidden = [t.view(torch.float16)[:, 0::2] if self.acts_split else t for t in (*acts[0], *acts[1])]  # (the hi halves carry the magnitude)
            m_ = torch.stack([t.abs().max().float() for t in (x0, x1, *hidden)]).max()
            self.domain_max = m_ if self.domain_max is None else torch.maximum(self.domain_max, m_)
        return ys, acts

    def pack_training(self, with_backward: bool) -> None:
        """The forward streams of both networks and (with_backward) their transposed streams for the fused backward pass, in ONE
        launch (`lt_mlp_pack_training`): a training step re-packs everything, the optimizer has moved the weights."""
        lib = _abi.load()
        vp = ctypes.c_void_p
        args = []
        for net in (self.a, self.b):
            if with_backward and getattr(net, "bpacked", None) is None:
                n = ctypes.c_size_t()
                _abi.check(lib.lt_mlp_backward_packed_floats(ctypes.byref(net.desc), ctypes.byref(n)), "lt_mlp_backward_packed_floats")
                net.bpacked = torch.zeros(int(n.value), device=net.packed.device, dtype=torch.float32)
            L = len(net.linears)
            arr = ctypes.c_void_p * L
            args += [ctypes.byref(net.desc), arr(*[l.weight.data_ptr() for l in net.linears]), arr(*[l.bias.data_ptr() for l in net.linears]),
                     vp(net.packed.data_ptr()), vp(net.bpacked.data_ptr()) if with_backward else vp(None)]
        _abi.check(lib.lt_mlp_pack_training(*args, PackedMLP._stream()), "lt_mlp_pack_training")
        self._bpacked_fresh = bool(with_backward)

    def split_rows(self, x: torch.Tensor) -> torch.Tensor:
        """Observation rows in the split format (lt_split_rows): converted once per PPO update for the first layer's weight gradient.
        A width that is not a multiple of 4 (the locomotion task's 270) is padded with zero columns to the next one: lt_wgrad's
        16-byte operand pieces need it, and the padded columns of dW are dropped again (`_backward_fused`)."""
        lib = _abi.load()
        k = x.shape[1]
        if k % 4:
            xp = torch.zeros(x.shape[0], k + (-k) % 4, device=x.device, dtype=torch.float32)
            xp[:, :k] = x
            x = xp
        out = torch.empty_like(x, dtype=torch.float32)
        _abi.check(lib.lt_split_rows(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), x.numel(), PackedMLP._stream()), "lt_split_rows")
        return out

    def _fused_backward_possible(self, x0, x1) -> bool:
        from .linear import _head_wgrad_ok

        if not (USE_FUSED_BACKWARD and USE_SPLIT_F16_WGRAD):
            return False
        if x0.dtype != torch.float32 or x1.dtype != torch.float32 or not x0.is_contiguous() or not x1.is_contiguous():
            return False
        if getattr(self, "_bwd_shapes_ok", None) is None:  # a property of the two networks: decided once
            ok = self.a.backward_ok() and self.b.backward_ok()
            for net in (self.a, self.b):
                dims = [net.desc.dims[i] for i in range(net.desc.num_layers + 1)]
                ok = ok and dims[0] % 2 == 0  # (2 mod 4: padded by split_rows)
                ok = ok and _head_wgrad_ok(torch.empty(0, dims[-1], device=x0.device), torch.empty(0, dims[-2], device=x0.device))
            self._bwd_shapes_ok = bool(ok)
        return self._bwd_shapes_ok

    def _fused_backward_ok(self, x0, x1, dy0, dy1) -> bool:
        return self._fused_backward_possible(x0, x1) and dy0.dtype == torch.float32 and dy1.dtype == torch.float32

    def _backward_fused(self, x0, x1, acts, dy0, dy1, grad_of, sums, x_split_rows=None, after_first=None, dy_amax=None) -> None:
        """The backward pass as: two head launches (lt_head_wgrad), ONE launch for both stacks' chains of input gradients
        (lt_mlp_backward_pair: dz of every hidden layer, ELU' applied in the layer epilogue, per-workgroup max |dz|), six weight
        gradients on the matrix cores that also leave the bias gradients' partials (lt_wgrad), one launch of ordered sums."""
        lib = _abi.load()
        vp = ctypes.c_void_p
        stream = PackedMLP._stream()
        m = x0.shape[0]
        nets, xs = (self.a, self.b), (x0, x1)
        asp = int(getattr(self, "acts_split", False))   # format of `acts` (forward_raw)
        xsp = int(x_split_rows is not None)             # the observation rows in the split format, from the caller (once per update)
        if not xsp and any(x.shape[1] % 4 for x in xs):  # widths 2 mod 4 reach lt_wgrad padded and split only
            x_split_rows, xsp = (self.split_rows(x0), self.split_rows(x1)), 1
        if xsp:
            xs = x_split_rows
        dys = [d if d.is_contiguous() else d.contiguous() for d in (dy0, dy1)]
        dev = x0.device
        if getattr(self, "sat", None) is None:
            self.sat = torch.zeros(1, device=dev, dtype=torch.float32)  # workgroups x layers of the chain that saturated (domain check)
        nblk = int(lib.lt_mlp_backward_blocks(ctypes.byref(nets[0].desc), ctypes.byref(nets[1].desc), m))
        dzs, amaxs, arrs = [], [], []
        # one list of ordered sums per network when the caller wants the first network's gradients early (`after_first`: the trainer
        # starts the all-reduce of the actor's half of the bucket under the critic's weight gradients), else one list for both
        per_net = (SumJobs(), SumJobs()) if after_first is not None else (sums, sums)
        for k, net in enumerate(nets):
            if not getattr(self, "_bpacked_fresh", False):  # (forward_raw packs the transposed streams with the forward ones)
                net.pack_backward()
            L = len(net.linears)
            n, kk = dys[k].shape[1], acts[k][L - 2].shape[1]
            ws = torch.empty(int(lib.lt_head_wgrad_ws_floats(m, n, kk)), device=dev, dtype=torch.float32)
            _abi.check(lib.lt_head_wgrad(vp(dys[k].data_ptr()), vp(acts[k][L - 2].data_ptr()), asp, m, n, kk, vp(None), vp(None), vp(ws.data_ptr()), stream), "lt_head_wgrad")
            per_net[k].add(ws, int(lib.lt_head_wgrad_nblk(m)), n * kk + 16, n * kk + n, n * kk, grad_of[net.linears[L - 1].weight], grad_of[net.linears[L - 1].bias])
            dz = [torch.empty_like(a) for a in acts[k]]
            am = torch.empty(L - 1, nblk, device=dev, dtype=torch.float32)
            dzs.append(dz)
            amaxs.append(am)
            arr = ctypes.c_void_p * (L - 1)
            arrs.append((arr(*[a.data_ptr() for a in acts[k]]), arr(*[t.data_ptr() for t in dz]), arr(*[am[l].data_ptr() for l in range(L - 1)])))
        # `dy_amax`: (max |dy0|, max |dy1|) as device scalars (lt_ppo_loss leaves them): the chain then runs on ONE scale per network
        # and writes every dz in the split format, scaled - what lt_wgrad reads without converting
        dzsp = int(dy_amax is not None and USE_SPLIT_DZ)
        scales = torch.empty(2, device=dev, dtype=torch.float32)
        _abi.check(lib.lt_mlp_backward_pair(ctypes.byref(nets[0].desc), vp(nets[0].bpacked.data_ptr()), vp(dys[0].data_ptr()), *arrs[0],
                                            ctypes.byref(nets[1].desc), vp(nets[1].bpacked.data_ptr()), vp(dys[1].data_ptr()), *arrs[1],
                                            m, asp, vp(dy_amax[0].data_ptr()) if dzsp else vp(None), vp(dy_amax[1].data_ptr()) if dzsp else vp(None),
                                            dzsp, vp(scales.data_ptr()), vp(self.sat.data_ptr()), stream), "lt_mlp_backward_pair")
        for k, net in enumerate(nets):
            for l in range(len(net.linears) - 2, -1, -1):
                inp = acts[k][l - 1] if l > 0 else xs[k]
                dz = dzs[k][l]
                n, kk = dz.shape[1], inp.shape[1]
                sp = int(lib.lt_wgrad_splits(m, n, kk))
                slabs = torch.empty(sp * n * kk + sp * n, device=dev, dtype=torch.float32)
                dbs = slabs[sp * n * kk:]
                _abi.check(lib.lt_wgrad(vp(dz.data_ptr()), dzsp, vp(scales[k:k + 1].data_ptr()), vp(inp.data_ptr()), asp if l > 0 else xsp, m, n, kk,
                                        vp(amaxs[k][l].data_ptr()), nblk, vp(slabs.data_ptr()), vp(dbs.data_ptr()), stream), "lt_wgrad")
                gw = grad_of[net.linears[l].weight]
                if kk != gw.shape[1]:  # padded first layer: the sum lands in a [n][kk] scratch, its first columns are the gradient
                    pad = torch.empty(n, kk, device=dev, dtype=torch.float32)
                    per_net[k].add(slabs, sp, n * kk, n * kk, n * kk, pad)
                    per_net[k].after.append((gw, pad[:, :gw.shape[1]]))
                else:
                    per_net[k].add(slabs, sp, n * kk, n * kk, n * kk, gw)
                per_net[k].add(dbs, sp, n, n, n, grad_of[net.linears[l].bias])
            if after_first is not None:
                per_net[k].launch()
                if k == 0:
                    after_first()
        self._keep_bwd = (dzs, amaxs, dys, per_net, scales if dzsp else None)
        self._bpacked_fresh = False  # the optimizer steps next

    def saturated(self) -> torch.Tensor | None:
        """Device counter of saturated workgroups of the fused backward chain (None: that path has not run)."""
        return getattr(self, "sat", None)

    def backward_raw(self, x0, x1, acts, dy0, dy1, grad_of, x_split_rows=None, after_first=None, dy_amax=None) -> None:
        """Both stacks' backward passes, every parameter gradient written into `grad_of[param]` (the flat bucket's views).
        `x_split_rows`: (x0, x1) in the split format (`split_rows`), if the caller has them.  `dy_amax`: device scalars (max |dy0|,
        max |dy1|), if the producer of the gradients left them (lt_ppo_loss: acc[20], acc[21]) - the hidden-layer gradients then travel
        in the split format as well.  `after_first()`: called when every
        gradient of the FIRST stack (the actor) has been enqueued - before the second stack's weight gradients are."""
        sums = SumJobs()
        if self._fused_backward_ok(x0, x1, dy0, dy1):
            self._backward_fused(x0, x1, acts, dy0, dy1, grad_of, sums, x_split_rows, after_first, dy_amax)
            sums.launch()
            return
        if getattr(self, "acts_split", False):
            raise RuntimeError("PackedPair.backward_raw: the activations are in the split format, which only the fused backward pass reads "
                               "(forward_raw(..., split=False) for the library path)")
        for net, x, a, dy in ((self.a, x0, acts[0], dy0), (self.b, x1, acts[1], dy1)):
            ws = [lin.weight for lin in net.linears]
            backward_chain(ws, [grad_of[lin.bias] for lin in net.linears], [grad_of[lin.weight] for lin in net.linears], x, a, dy, sums)
        sums.launch()  # every ordered sum of partials of both stacks: one launch

    def __call__(self, x0: torch.Tensor, x1: torch.Tensor):
        self.a.pack()
        self.b.pack()
        params = []
        for net in (self.a, self.b):
            for lin in net.linears:
                params += [lin.weight, lin.bias]
        return _PairForward.apply(self, x0.contiguous(), x1.contiguous(), *params)
