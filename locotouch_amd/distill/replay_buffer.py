"""Trajectory replay buffer of the DAgger loop (reference locotouch/distill/replay_buffer.py:9-160), laid out for the device.

Same contract: `collect_data(teacher_policy, student_policy, num_steps)` rolls the env with the teacher (first iteration) or
the student (later ones) and keeps every trajectory that FINISHES during the call, in order of completion (env-id order inside
a step), until the kept steps reach `num_steps`; `to_recurrent_generator(batch_size)` yields padded (L, B, ...) batches of whole
trajectories in a random order; `evaluate(student, num_trajs)` returns trajectory rewards / lengths.

What is different is where the data lives and when the host looks at it:
  * every step of a collection is written into device tensors [T][N][D] (policy rows = proprioception | object state, the
    DELAYED tactile rows, rewards, dones); a trajectory is three integers (first row, length, stride) - the reference stacks
    per-env Python lists of per-step tensors (O(T) tiny kernels per finished trajectory) and keeps one tensor per trajectory;
  * the env loop has NO per-step host sync: hidden-state / delay-line resets are masked arithmetic, and the trajectory
    bookkeeping (which needs `dones` on the host) runs every `check_every` steps on one small copy.  The loop therefore
    overshoots the reference's stopping step by at most `check_every - 1` env steps; those steps are discarded, so the kept
    trajectories, their order, and the returned rewards / lengths are what the reference's per-step bookkeeping gives;
  * a padded batch is ONE gather per field (row index grid built on the device) instead of a Python loop over trajectories.
"""
from __future__ import annotations

import numpy as np
import torch

from .tactile_recorder import TactileRecorder


class _Block:
    """One collection's step store, grown in chunks of `chunk` steps."""

    def __init__(self, n: int, dims: dict, device, chunk: int = 256):
        self.n, self.dims, self.device, self.chunk = n, dims, device, chunk
        self.chunks: dict[str, list[torch.Tensor]] = {k: [] for k in dims}
        self.steps = 0

    def row(self, key: str, t: int) -> torch.Tensor:
        c, i = divmod(t, self.chunk)
        lst = self.chunks[key]
        while len(lst) <= c:
            d, dtype = self.dims[key]
            lst.append(torch.zeros((self.chunk, self.n) + ((d,) if d else ()), dtype=dtype, device=self.device))
        return lst[c][i]

    def stacked(self, key: str, t0: int, t1: int) -> torch.Tensor:
        return torch.cat(self.chunks[key], dim=0)[t0:t1] if self.chunks[key] else torch.zeros(0, device=self.device)


class ReplayBuffer:
    def __init__(self, env, tactile_recorder: TactileRecorder, proprioception_dim: int, check_every: int = 16):
        self._env = env
        self._num_envs = env.num_envs
        self._device = env.device
        self._proprioception_dim = proprioception_dim
        self._tactile_recorder = tactile_recorder
        self._check_every = max(1, int(check_every))
        self._steps_count = 0
        self._reward_sums = torch.zeros(self._num_envs, device=self._device)
        # persistent store: flat rows [(sum of kept block steps) * N][D]; trajectories index into it
        self._policy_blocks: list[torch.Tensor] = []
        self._tactile_blocks: list[torch.Tensor] = []
        self._block_base: list[int] = []
        self._rows_total = 0
        self._traj_first: list[int] = []   # flat row of the trajectory's first step
        self._traj_len: list[int] = []
        self._flat = None                  # (policy rows, tactile rows) concatenated over blocks, built lazily
        self._traj_dev = None

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _split_obs(ret):
        obs, extras = ret
        return obs, extras["observations"]["tactile"]

    def collect_data(self, teacher_policy, student_policy, num_steps: int):
        env, n, rec = self._env, self._num_envs, self._tactile_recorder
        if student_policy is not None:
            env.reset()          # replay_buffer.py:22-23
            student_policy.reset()  # (the reference carries the previous collection's hidden state into the fresh episodes)
        rec.reset()
        obs, tactile = self._split_obs(env.get_observations())
        dims = {"policy": (obs.shape[1], torch.float32), "tactile": (tactile.shape[1], torch.float32),
                "reward": (0, torch.float32), "dones": (0, torch.bool)}
        blk = _Block(n, dims, self._device)
        start_idx = np.zeros(n, dtype=np.int64)
        start_count = self._steps_count
        rewards_out, lengths_out, trajs = [], [], []   # trajs: (env, start, end) of this block
        reward_sums = self._reward_sums.cpu().numpy().astype(np.float64)
        t, t_seen, stop_t = 0, 0, None

        def bookkeeping(upto: int):
            """The reference's per-step trajectory bookkeeping (replay_buffer.py:58-72) for steps t_seen .. upto-1."""
            nonlocal t_seen, stop_t
            if upto <= t_seen:
                return
            d = blk.stacked("dones", t_seen, upto).cpu().numpy()
            r = blk.stacked("reward", t_seen, upto).cpu().numpy()
            for i in range(upto - t_seen):
                steps_count = t_seen + i + 1
                reward_sums[:] += r[i]
                done_idx = np.nonzero(d[i])[0]
                if done_idx.size:
                    rewards_out.extend(reward_sums[done_idx].tolist())
                    lengths_out.extend((steps_count - start_idx[done_idx]).tolist())
                    reward_sums[done_idx] = 0
                    for e in done_idx:
                        if self._steps_count - start_count < num_steps:
                            if start_idx[e] < steps_count:
                                self._steps_count += int(steps_count - start_idx[e])
                                trajs.append((int(e), int(start_idx[e]), steps_count))
                            start_idx[e] = steps_count
                        else:
                            break
                if not (self._steps_count - start_count < num_steps):
                    stop_t = steps_count
                    break
            t_seen = upto

        with torch.no_grad():
            while stop_t is None:
                proprio = obs[:, :self._proprioception_dim]
                action = teacher_policy(obs) if student_policy is None else student_policy(proprio, tactile)
                blk.row("policy", t).copy_(obs)          # before the step overwrites the env's rows
                rec.record_new_tactile_signals(tactile)
                blk.row("tactile", t).copy_(rec.get_tactile_signals())
                obs, reward, dones, extras = env.step(action)
                tactile = extras["observations"]["tactile"]
                blk.row("reward", t).copy_(reward)
                done_mask = dones != 0
                blk.row("dones", t).copy_(done_mask)
                if student_policy is not None:
                    student_policy.reset(done_mask)
                rec.reset(done_mask)
                t += 1
                if t % self._check_every == 0:
                    bookkeeping(t)
        self._reward_sums.copy_(torch.from_numpy(reward_sums.astype(np.float32)))
        # keep the block's rows up to the stopping step
        keep = stop_t
        base = self._rows_total
        self._policy_blocks.append(blk.stacked("policy", 0, keep).reshape(keep * n, -1))
        self._tactile_blocks.append(blk.stacked("tactile", 0, keep).reshape(keep * n, -1))
        self._block_base.append(base)
        self._rows_total += keep * n
        for e, s, end in trajs:
            self._traj_first.append(base + s * n + e)
            self._traj_len.append(end - s)
        self._flat = self._traj_dev = None
        return rewards_out, lengths_out

    # ------------------------------------------------------------------------------------------------------------
    def _materialise(self):
        if self._flat is None:
            self._flat = (torch.cat(self._policy_blocks, dim=0), torch.cat(self._tactile_blocks, dim=0))
            self._policy_blocks, self._tactile_blocks = [self._flat[0]], [self._flat[1]]  # one copy only
            self._traj_dev = (torch.tensor(self._traj_first, dtype=torch.int64, device=self._device),
                              torch.tensor(self._traj_len, dtype=torch.int64, device=self._device))
        return self._flat, self._traj_dev

    def to_recurrent_generator(self, batch_size: int, static_shapes: bool = True):
        """`static_shapes`: every batch is padded to the SAME (longest trajectory in the buffer, batch_size) shape - masked
        rows / columns, so losses and gradients are those of the tight padding.  On ROCm every new (L * B) costs MIOpen a
        solver search for the student's convolutions (110-300 ms per new shape, tools/shape_probe.py; once 151 s for a
        kernel compile), which at ~10 differently shaped batches per epoch was 85 % of the BC wall clock."""
        num_trajs = len(self._traj_first)
        order = np.random.permutation(np.arange(num_trajs))  # replay_buffer.py:86-87
        pad = (max(self._traj_len), batch_size) if static_shapes and num_trajs else None
        for s in range(0, num_trajs, batch_size):
            yield self._prepare_padded_sequence(order[s:min(s + batch_size, num_trajs)], pad_to=pad)

    def _prepare_padded_sequence(self, traj_indices, pad_to=None):
        (policy, tactile), (first, length) = self._materialise()
        idx = torch.as_tensor(np.asarray(traj_indices), dtype=torch.int64, device=self._device)
        f, ln = first[idx], length[idx]
        max_len = int(max(self._traj_len[i] for i in traj_indices))
        if pad_to is not None:
            max_len = max(max_len, int(pad_to[0]))
            extra = int(pad_to[1]) - len(traj_indices)
            if extra > 0:  # empty trajectories: length 0, every step masked
                f = torch.cat([f, f.new_zeros(extra)])
                ln = torch.cat([ln, ln.new_zeros(extra)])
        tt = torch.arange(max_len, device=self._device).unsqueeze(1)            # (L, 1)
        masks = tt < ln.unsqueeze(0)                                            # (L, B)
        rows = torch.where(masks, f.unsqueeze(0) + tt * self._num_envs, torch.zeros_like(tt))
        m = masks.unsqueeze(-1)
        pol = policy[rows] * m
        return dict(proprioceptions=pol[..., :self._proprioception_dim], teacher_encoder_obses=pol[..., self._proprioception_dim:],
                    tactile_signals=tactile[rows] * m, masks=masks)

    def clear_buffer(self):
        self._policy_blocks, self._tactile_blocks, self._block_base = [], [], []
        self._traj_first, self._traj_len = [], []
        self._rows_total = 0
        self._flat = self._traj_dev = None
        self._steps_count = 0
        self._reward_sums[:] = 0

    def evaluate(self, student_policy, num_trajs: int):
        """replay_buffer.py:131-151: roll the student until `num_trajs` episodes have finished (from the env's current state)."""
        env, n = self._env, self._num_envs
        rewards, lengths = [], []
        reward_sums = self._reward_sums.cpu().numpy().astype(np.float64)
        env_steps = np.zeros(n, dtype=np.int64)
        k = self._check_every
        rbuf = torch.zeros(k, n, device=self._device)
        dbuf = torch.zeros(k, n, dtype=torch.bool, device=self._device)
        with torch.no_grad():
            obs, tactile = self._split_obs(env.get_observations())
            while len(rewards) < num_trajs:
                for i in range(k):
                    action = student_policy(obs[:, :self._proprioception_dim], tactile)
                    obs, reward, dones, extras = env.step(action)
                    tactile = extras["observations"]["tactile"]
                    rbuf[i].copy_(reward)
                    dbuf[i].copy_(dones != 0)
                    student_policy.reset(dbuf[i])  # (the reference lets the GRU state leak into the next episode here)
                r, d = rbuf.cpu().numpy(), dbuf.cpu().numpy()
                for i in range(k):
                    reward_sums += r[i]
                    env_steps += 1
                    done_idx = np.nonzero(d[i])[0]
                    if done_idx.size:
                        rewards.extend(reward_sums[done_idx].tolist())
                        lengths.extend(env_steps[done_idx].astype(np.float64).tolist())
                        reward_sums[done_idx] = 0
                        env_steps[done_idx] = 0
                    if len(rewards) >= num_trajs:
                        break
        self._reward_sums.copy_(torch.from_numpy(reward_sums.astype(np.float32)))
        return rewards, lengths

    @property
    def num_trajs(self) -> int:
        return len(self._traj_first)

    @property
    def num_steps(self) -> int:
        return self._steps_count
