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

    def forward(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if x.dim() != 2 or x.shape[1] != self.in_features or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("x must be a contiguous float32 [m, in_features] tensor")
        m = x.shape[0]
        if out is None:
            out = torch.empty(m, self.out_features, device=x.device, dtype=torch.float32)
        _abi.check(self.lib.lt_mlp_forward(ctypes.byref(self.desc), ctypes.c_void_p(self.packed.data_ptr()), ctypes.c_void_p(x.data_ptr()), m,
                                           ctypes.c_void_p(out.data_ptr()), self._stream()), "lt_mlp_forward")
        return out

    __call__ = forward
