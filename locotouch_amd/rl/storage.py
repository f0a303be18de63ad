"""Rollout buffer + GAE(lambda) + flat minibatch sampling for PPO.

Same contract as the reference's `RolloutStorage` (loco_rl/loco_rl/storage/rollout_storage.py:13-243): (T, N, .)
buffers, `compute_returns` = GAE with the global (unbiased-std) advantage normalisation, and a minibatch generator
that draws ONE `randperm` per update and reuses it for every epoch (:189).  In a multi-GPU job each rank holds its own
env shard; the advantage statistics are then all-reduced so they equal the single-process statistics over all shards
(SURVEY.md §8(e) item 3).
"""
from __future__ import annotations

from typing import Iterator, NamedTuple

import torch

from .dist import Dist


class Batch(NamedTuple):
    obs: torch.Tensor
    critic_obs: torch.Tensor
    actions: torch.Tensor
    values: torch.Tensor
    advantages: torch.Tensor
    returns: torch.Tensor
    log_prob: torch.Tensor
    mu: torch.Tensor
    sigma: torch.Tensor


class RolloutStorage:
    def __init__(self, num_envs: int, num_steps: int, obs_dim: int, critic_obs_dim: int, num_actions: int, device="cpu",
                 obs_dtype: torch.dtype = torch.float32):
        """`obs_dtype`: torch.float32 (the reference's, rollout_storage.py:36-44) or torch.bfloat16 (BASELINE config 5: the env
        kernel writes bf16 rows straight into the slots, the policy kernel reads them; the update widens its minibatches to f32)."""
        self.num_envs, self.num_steps, self.device = num_envs, num_steps, device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, device=device, dtype=dtype)  # noqa: E731
        self.observations = z(num_steps, num_envs, obs_dim, dtype=obs_dtype)
        self.privileged_observations = z(num_steps, num_envs, critic_obs_dim, dtype=obs_dtype)
        self.actions = z(num_steps, num_envs, num_actions)
        self.mu = z(num_steps, num_envs, num_actions)
        self.sigma = z(num_steps, num_envs, num_actions)
        self.rewards = z(num_steps, num_envs, 1)
        self.dones = z(num_steps, num_envs, 1, dtype=torch.uint8)
        self.values = z(num_steps, num_envs, 1)
        self.returns = z(num_steps, num_envs, 1)
        self.advantages = z(num_steps, num_envs, 1)
        self.actions_log_prob = z(num_steps, num_envs, 1)
        self.saved_hidden_states_a = self.saved_hidden_states_c = None  # recurrent policies: lists of (T, layers, N, H)
        self.step = 0

    def add(self, obs, critic_obs, actions, rewards, dones, values, log_prob, mu, sigma, hidden_states=None) -> None:
        t = self.step
        if t >= self.num_steps:
            raise OverflowError("rollout buffer is full: call clear() before adding transitions")
        if hidden_states is not None and hidden_states != (None, None):  # rollout_storage.py:109-146
            hid_a = hidden_states[0] if isinstance(hidden_states[0], tuple) else (hidden_states[0],)
            hid_c = None if hidden_states[1] is None else (hidden_states[1] if isinstance(hidden_states[1], tuple) else (hidden_states[1],))
            if self.saved_hidden_states_a is None:
                self.saved_hidden_states_a = [torch.zeros(self.num_steps, *h.shape, device=self.device) for h in hid_a]
                if hid_c is not None:
                    self.saved_hidden_states_c = [torch.zeros(self.num_steps, *h.shape, device=self.device) for h in hid_c]
            for i, h in enumerate(hid_a):
                self.saved_hidden_states_a[i][t].copy_(h)
            if hid_c is not None:
                for i, h in enumerate(hid_c):
                    self.saved_hidden_states_c[i][t].copy_(h)
        self.observations[t].copy_(obs)
        self.privileged_observations[t].copy_(critic_obs)
        self.actions[t].copy_(actions)
        self.rewards[t].copy_(rewards.view(-1, 1))
        self.dones[t].copy_(dones.view(-1, 1))
        self.values[t].copy_(values)
        self.actions_log_prob[t].copy_(log_prob.view(-1, 1))
        self.mu[t].copy_(mu)
        self.sigma[t].copy_(sigma)
        self.step = t + 1

    def clear(self) -> None:
        self.step = 0

    def compute_returns(self, last_values: torch.Tensor, gamma: float, lam: float, normalize_advantage: bool = True,
                        dist: Dist | None = None) -> None:
        """GAE: delta_t = r_t + gamma (1-d_t) V_{t+1} - V_t ;  A_t = delta_t + gamma lam (1-d_t) A_{t+1} ; R_t = A_t + V_t."""
        if self.rewards.is_cuda:  # one launch instead of ~7 tensor ops per step (csrc/lt_ppo.hip lt_gae)
            import ctypes

            from .. import _abi

            vp = ctypes.c_void_p
            lv = last_values.detach().reshape(-1).contiguous().float()
            if self.advantages.shape != self.returns.shape or not self.advantages.is_contiguous():
                self.advantages = torch.empty_like(self.returns)
            _abi.check(_abi.load().lt_gae(vp(self.rewards.data_ptr()), vp(self.dones.data_ptr()), vp(self.values.data_ptr()), vp(lv.data_ptr()),
                                          float(gamma), float(lam), self.num_steps, self.num_envs, vp(self.returns.data_ptr()),
                                          vp(self.advantages.data_ptr()), vp(torch.cuda.current_stream(self.rewards.device).cuda_stream)), "lt_gae")
        else:
            gae = 0
            next_values = last_values
            for t in range(self.num_steps - 1, -1, -1):
                alive = 1.0 - self.dones[t].float()
                delta = self.rewards[t] + alive * gamma * next_values - self.values[t]
                gae = delta + alive * gamma * lam * gae
                self.returns[t] = gae + self.values[t]
                next_values = self.values[t]
            self.advantages = self.returns - self.values
        if normalize_advantage:
            if dist is None or dist.world_size == 1:
                self.advantages = (self.advantages - self.advantages.mean()) / (self.advantages.std() + 1e-8)
            else:
                a = self.advantages
                stats = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), device=a.device)]).double()
                dist.all_reduce_sum_(stats)
                n = stats[2]
                mean = stats[0] / n
                var = (stats[1] - n * mean * mean) / (n - 1.0)  # unbiased, like torch.std
                self.advantages = (a - mean.float()) / (var.clamp_min(0).sqrt().float() + 1e-8)

    def mini_batches(self, num_mini_batches: int, num_epochs: int) -> Iterator[Batch]:
        total = self.num_envs * self.num_steps
        mb = total // num_mini_batches
        perm = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        flat = [x.flatten(0, 1) for x in (self.observations, self.privileged_observations, self.actions, self.values,
                                          self.advantages, self.returns, self.actions_log_prob, self.mu, self.sigma)]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = perm[i * mb:(i + 1) * mb]
                b = [x[idx] for x in flat]
                if b[0].dtype != torch.float32:  # bf16 observation storage: the update computes in f32
                    b[0], b[1] = b[0].float(), b[1].float()
                yield Batch(*b)

    def mini_batch_indices(self, num_mini_batches: int, num_epochs: int) -> Iterator[torch.Tensor]:
        """The row indices `mini_batches` gathers with (same single `randperm`, same order), for consumers that gather themselves
        (the fused PPO loss reads the small per-row tensors through the index, csrc/lt_ppo.hip)."""
        perm, mb = self.mini_batch_permutation(num_mini_batches)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                yield perm[i * mb:(i + 1) * mb]

    def mini_batch_permutation(self, num_mini_batches: int):
        """(perm, rows per minibatch): the ONE permutation an update draws and reuses in every epoch (rollout_storage.py:189-190);
        minibatch i is perm[i * mb:(i + 1) * mb]."""
        mb = (self.num_envs * self.num_steps) // num_mini_batches
        return torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device), mb

    def recurrent_mini_batches(self, num_mini_batches: int, num_epochs: int = 8, hidden_states_a="saved", hidden_states_c="saved"):
        """Minibatches of whole trajectories for recurrent policies (reference rollout_storage.py:246-318): envs are dealt to
        minibatches in contiguous blocks; a block's padded trajectories (and the hidden states saved at their first steps)
        form the batch.  `hidden_states_*`: lists of (T, layers, N, H) tensors saved during the rollout, or None.
        Yields (obs, critic_obs, actions, values, advantages, returns, log_prob, mu, sigma, (hid_a, hid_c), masks)."""
        from .trajectories import split_and_pad_trajectories

        if isinstance(hidden_states_a, str):
            hidden_states_a = self.saved_hidden_states_a
        if isinstance(hidden_states_c, str):
            hidden_states_c = self.saved_hidden_states_c
        obs_traj, masks = split_and_pad_trajectories(self.observations, self.dones)
        critic_traj, _ = split_and_pad_trajectories(self.privileged_observations, self.dones)
        per = self.num_envs // num_mini_batches
        d = self.dones.squeeze(-1).bool()
        starts = torch.ones_like(d)          # a trajectory starts at t = 0 and right after every done
        starts[1:] = d[:-1]
        traj_per_env = starts.sum(0)         # (N,)
        first_of_env = torch.cumsum(traj_per_env, 0) - traj_per_env
        starts_env_major = starts.t()        # (N, T)

        def hidden(saved, lo, hi):
            if saved is None:
                return None
            out = [h.permute(2, 0, 1, 3)[starts_env_major][lo:hi].transpose(1, 0).contiguous() for h in saved]
            return out[0] if len(out) == 1 else out

        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                e0, e1 = i * per, (i + 1) * per
                lo = int(first_of_env[e0])
                hi = lo + int(traj_per_env[e0:e1].sum())
                sl = (slice(None), slice(e0, e1))
                yield (obs_traj[:, lo:hi], critic_traj[:, lo:hi], self.actions[sl], self.values[sl], self.advantages[sl],
                       self.returns[sl], self.actions_log_prob[sl], self.mu[sl], self.sigma[sl],
                       (hidden(hidden_states_a, lo, hi), hidden(hidden_states_c, lo, hi)), masks[:, lo:hi])

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        ends = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (ends[1:] - ends[:-1]).float().mean(), self.rewards.mean()
