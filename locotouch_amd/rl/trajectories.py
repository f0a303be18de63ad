"""Trajectory splitting / padding for recurrent policies and the distillation buffers.

Behavioural twins of the reference's `split_and_pad_trajectories` / `unpad_trajectories`
(loco_rl/loco_rl/utils/utils.py:37-83): a (T, N, D) rollout is cut at the `dones` (and at the end of the buffer), the
pieces are laid out env-major as columns of a (T, n_traj, D) tensor, zero-padded, with a boolean (T, n_traj) validity mask.
"""
from __future__ import annotations

import torch


def trajectory_lengths(dones: torch.Tensor) -> torch.Tensor:
    """Lengths of the pieces, env-major, for dones of shape (T, N, 1) or (T, N); the buffer end closes every env's last piece."""
    d = dones.reshape(dones.shape[0], dones.shape[1]).clone().bool()
    d[-1] = True
    ends = d.t().reshape(-1).nonzero(as_tuple=False).flatten()       # flat index = env * T + t
    starts = torch.cat([ends.new_full((1,), -1), ends[:-1]])
    return ends - starts


def split_and_pad_trajectories(tensor: torch.Tensor, dones: torch.Tensor):
    T = tensor.shape[0]
    lengths = trajectory_lengths(dones)
    n_traj = int(lengths.numel())
    flat = tensor.transpose(0, 1).reshape(-1, tensor.shape[-1])      # env-major rows
    # position of every row inside its piece, and the piece it belongs to
    piece = torch.repeat_interleave(torch.arange(n_traj, device=tensor.device), lengths)
    first = torch.cumsum(lengths, 0) - lengths
    pos = torch.arange(flat.shape[0], device=tensor.device) - first[piece]
    padded = tensor.new_zeros(T, n_traj, tensor.shape[-1])
    padded[pos, piece] = flat
    masks = lengths.unsqueeze(0) > torch.arange(T, device=tensor.device).unsqueeze(1)
    return padded, masks


def unpad_trajectories(trajectories: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """Inverse of split_and_pad_trajectories: back to (T, N, D)."""
    T = trajectories.shape[0]
    rows = trajectories.transpose(0, 1)[masks.transpose(0, 1)]        # env-major rows, padding dropped
    return rows.view(-1, T, trajectories.shape[-1]).transpose(0, 1)
