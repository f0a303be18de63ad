"""Running observation normaliser of the reference's runner (loco_rl/loco_rl/modules/normalizer.py:14-76, used at
loco_rl/loco_rl/runners/on_policy_runner.py:85-95 when `empirical_normalization` is set; off in every LocoTouch agent cfg).

y = (x - mean) / (std + eps) with mean / var tracked by the parallel (Chan et al.) update over batches while in training
mode and until `until` samples have been seen.  Buffer names (`_mean`, `_var`, `_std`, `count`) are the reference's, so the
`obs_norm_state_dict` / `critic_obs_norm_state_dict` entries of a checkpoint interchange.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class EmpiricalNormalization(nn.Module):
    def __init__(self, shape, eps: float = 1e-2, until: int | None = None):
        super().__init__()
        self.eps, self.until = eps, until
        self.register_buffer("_mean", torch.zeros(shape).unsqueeze(0))
        self.register_buffer("_var", torch.ones(shape).unsqueeze(0))
        self.register_buffer("_std", torch.ones(shape).unsqueeze(0))
        self.register_buffer("count", torch.tensor(0, dtype=torch.long))

    @property
    def mean(self) -> torch.Tensor:
        return self._mean.squeeze(0).clone()

    @property
    def std(self) -> torch.Tensor:
        return self._std.squeeze(0).clone()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            self.update(x)
        return (x - self._mean) / (self._std + self.eps)

    @torch.no_grad()
    def update(self, x: torch.Tensor) -> None:
        if self.until is not None and int(self.count) >= self.until:
            return
        n = x.shape[0]
        self.count += n
        w = n / float(self.count)                                   # weight of the new batch in the merged statistics
        batch_mean = x.mean(dim=0, keepdim=True)
        batch_var = x.var(dim=0, unbiased=False, keepdim=True)
        shift = batch_mean - self._mean
        self._mean += w * shift
        self._var += w * (batch_var - self._var + shift * (batch_mean - self._mean))
        self._std = self._var.sqrt()

    def inverse(self, y: torch.Tensor) -> torch.Tensor:
        return y * (self._std + self.eps) + self._mean
