"""Per-env tactile delay line (reference locotouch/distill/tactile_recorder.py:4-34): a `max_delay`-deep shift register of
the tactile rows and one delay index per env, redrawn at that env's reset; the first signal after a reset fills the whole
register.  Everything is masked arithmetic on the device - no `nonzero()` / host sync per step (the reference has two)."""
from __future__ import annotations

import torch


class TactileRecorder:
    def __init__(self, device, env_num: int, tactile_shape, min_delay: int = 3, max_delay: int = 7):
        self.device = device
        self.env_num = env_num
        self.tactile_shape = (tactile_shape,) if isinstance(tactile_shape, int) else tuple(tactile_shape)
        self.min_delay, self.max_delay = min_delay, max_delay
        self.tactile_buffer = torch.zeros((env_num, max_delay, *self.tactile_shape), dtype=torch.float32, device=device)
        self.first_signal_recorded = torch.ones(env_num, dtype=torch.bool, device=device)
        self.delay_steps = torch.zeros(env_num, dtype=torch.long, device=device)
        self.env_idx = torch.arange(env_num, device=device)
        self.reset()

    def _mask(self, env_idx) -> torch.Tensor:
        if env_idx is None:
            return torch.ones(self.env_num, dtype=torch.bool, device=self.device)
        if env_idx.dtype == torch.bool:
            return env_idx
        m = torch.zeros(self.env_num, dtype=torch.bool, device=self.device)
        m[env_idx] = True
        return m

    def reset(self, env_idx=None):
        """`env_idx`: index tensor (reference call form), bool mask [env_num], or None = all."""
        m = self._mask(env_idx)
        self.tactile_buffer.mul_((~m).to(torch.float32).view(-1, *([1] * (self.tactile_buffer.dim() - 1))))
        self.first_signal_recorded |= m
        # torch.randint(low=min_delay, high=max_delay): the upper bound is exclusive (tactile_recorder.py:22)
        fresh = torch.randint(low=self.min_delay, high=self.max_delay, size=(self.env_num,), device=self.device)
        self.delay_steps = torch.where(m, fresh, self.delay_steps)

    def record_new_tactile_signals(self, tactile_signals: torch.Tensor):
        shifted = torch.cat([tactile_signals.unsqueeze(1), self.tactile_buffer[:, :-1]], dim=1)
        filled = tactile_signals.unsqueeze(1).expand_as(self.tactile_buffer)
        first = self.first_signal_recorded.view(-1, *([1] * (self.tactile_buffer.dim() - 1)))
        self.tactile_buffer = torch.where(first, filled, shifted)
        self.first_signal_recorded = torch.zeros_like(self.first_signal_recorded)

    def get_tactile_signals(self) -> torch.Tensor:
        return self.tactile_buffer[self.env_idx, self.delay_steps]
