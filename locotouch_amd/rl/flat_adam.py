"""`clip_grad_norm_` + `Adam.step()` as two launches on flat buffers (csrc/lt_ppo.hip `lt_adam_clip_step`).

The reference ends every minibatch step with `nn.utils.clip_grad_norm_(...)` and `optimizer.step()`
(loco_rl/loco_rl/algorithms/ppo.py:318-319): on 17 small tensors that is ~12 multi-tensor launches, 160 us of a 2 ms step.
`FlatAdam.adopt(optimizer)` moves the parameters, their gradients and both Adam moments into four flat f32 buffers (the tensors
the module and the optimizer hold become VIEWS of them, so `state_dict()` / `load_state_dict()` of both keep the reference's
checkpoint layout) and `step()` runs the fused kernels.  Anything that re-binds those tensors (an `optimizer.load_state_dict`,
a `.to()`) is detected and re-adopted before the next step.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _abi


class FlatAdam:
    ALIGN = 64  # elements: every tensor starts on a 256-byte boundary (GEMM operands keep their alignment); the padding stays 0

    def __init__(self, optimizer: torch.optim.Adam):
        if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            raise TypeError("FlatAdam wraps a single-group torch.optim.Adam")
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            raise TypeError("FlatAdam: amsgrad / maximize / capturable / differentiable are not supported")
        self.optimizer = optimizer
        self.params = [p for p in g["params"] if p.requires_grad]
        if not self.params or any(p.dtype != torch.float32 or not p.is_cuda for p in self.params):
            raise TypeError("FlatAdam needs f32 CUDA parameters")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.n = off
        dev = self.params[0].device
        self.flat_p = torch.zeros(self.n, device=dev)
        self.flat_g = torch.zeros(self.n, device=dev)
        self.flat_m = torch.zeros(self.n, device=dev)
        self.flat_v = torch.zeros(self.n, device=dev)
        self._lib = _abi.load()
        self._ws = torch.empty(int(self._lib.lt_adam_clip_step_ws_floats(self.n)), device=dev)
        self.grad_norm = torch.zeros(1, device=dev)
        self._adopt()

    def _views(self, flat):
        for p, off in zip(self.params, self.offsets):
            yield p, flat[off:off + p.numel()].view_as(p)

    def _adopt(self) -> None:
        """Copy whatever the module / optimizer currently hold into the flat buffers and re-point them at views."""
        st = self.optimizer.state
        with torch.no_grad():
            for p, v in self._views(self.flat_p):
                if p.data.data_ptr() != v.data_ptr():
                    v.copy_(p.data)
                    p.data = v
            self._gviews = [v for _, v in self._views(self.flat_g)]
            steps = {float(st[p]["step"]) for p in self.params if p in st and "step" in st[p]}
            if len(steps) > 1:
                raise ValueError("FlatAdam: the parameters' Adam step counts differ")
            self.step_count = int(steps.pop()) if steps else 0
            for name, flat in (("exp_avg", self.flat_m), ("exp_avg_sq", self.flat_v)):
                for p, v in self._views(flat):
                    s = st[p]
                    if name not in s:
                        v.zero_()
                    elif s[name].data_ptr() != v.data_ptr():
                        v.copy_(s[name])
                    s[name] = v
            self._step_t = torch.tensor(float(self.step_count))  # the form torch's own (non-capturable) Adam keeps; ONE tensor, shared
            for p in self.params:
                st[p]["step"] = self._step_t

    def _bound(self) -> bool:
        p0 = self.params[0]
        s0 = self.optimizer.state.get(p0, {})
        return (p0.data.data_ptr() == self.flat_p.data_ptr() and "exp_avg" in s0 and s0["exp_avg"].data_ptr() == self.flat_m.data_ptr()
                and s0.get("step") is self._step_t)

    def zero_grad(self) -> None:
        """Gradients are set to None: backward then hands each parameter a fresh tensor (no per-parameter accumulate launches);
        `gather_grads` packs them into the flat bucket with one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def gather_grads(self) -> torch.Tensor:
        """Pack the parameters' gradients into `flat_g` (a missing gradient counts as zero) and re-point `.grad` at its views."""
        have = [(v, p.grad) for p, v in zip(self.params, self._gviews) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for p, v in zip(self.params, self._gviews) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if missing:
            torch._foreach_zero_(missing)
        for p, v in zip(self.params, self._gviews):
            p.grad = v
        return self.flat_g

    def grad_views(self) -> dict:
        """param -> its gradient's view in the flat bucket, `.grad` re-pointed at it: for a backward pass that writes the gradients
        in place (rl/ppo.py `_direct_update`), with no gather afterwards."""
        if not self._bound():
            self._adopt()
        for p, v in zip(self.params, self._gviews):
            p.grad = v
        return {p: v for p, v in zip(self.params, self._gviews)}

    def offset_of(self, param) -> int:
        """Element offset of `param`'s gradient inside the flat bucket (rl/ppo.py splits the bucket into per-network all-reduces)."""
        for p, off in zip(self.params, self.offsets):
            if p is param:
                return off
        raise KeyError("parameter not in the flat bucket")

    def step_dev(self, max_norm: float, lr_dev: torch.Tensor) -> None:
        """`step(gathered=True)` with the learning rate read from the device scalar `lr_dev` (lt_ppo_lr_rule keeps it)."""
        if not self._bound():
            self._adopt()
        g = self.optimizer.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        vp = ctypes.c_void_p
        _abi.check(self._lib.lt_adam_clip_step_dev(vp(self.flat_p.data_ptr()), vp(self.flat_g.data_ptr()), vp(self.flat_m.data_ptr()), vp(self.flat_v.data_ptr()),
                                                   self.n, float(max_norm or 0.0), vp(lr_dev.data_ptr()), float(b1), float(b2), float(g["eps"]),
                                                   float(g["weight_decay"]), self.step_count, vp(self._ws.data_ptr()), vp(self.grad_norm.data_ptr()),
                                                   vp(torch.cuda.current_stream(self.flat_p.device).cuda_stream)), "lt_adam_clip_step_dev")
        self._step_t.fill_(float(self.step_count))

    def step(self, max_norm: float, gathered: bool = False) -> None:
        if not self._bound():
            self._adopt()
        if not gathered:
            self.gather_grads()
        g = self.optimizer.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        vp = ctypes.c_void_p
        _abi.check(self._lib.lt_adam_clip_step(vp(self.flat_p.data_ptr()), vp(self.flat_g.data_ptr()), vp(self.flat_m.data_ptr()), vp(self.flat_v.data_ptr()),
                                               self.n, float(max_norm or 0.0), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                               float(g["weight_decay"]), self.step_count, vp(self._ws.data_ptr()), vp(self.grad_norm.data_ptr()),
                                               vp(torch.cuda.current_stream(self.flat_p.device).cuda_stream)), "lt_adam_clip_step")
        self._step_t.fill_(float(self.step_count))
