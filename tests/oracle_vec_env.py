"""Test-only VecEnv over the CPU oracle (torch CPU tensors viewing the oracle's host arena).

Lets CPU tests drive code that expects the VecEnv protocol (the trainer, the reference's unmodified launch scripts through
locotouch_amd.compat.runtime) without a GPU.  Never imported by the product path.
"""
from __future__ import annotations

import numpy as np
import torch

from locotouch_amd import _abi
from locotouch_amd.layout import Layout
from tests import oracle_lib

class OracleVecEnv:
    num_actions = 12

    def __init__(self, task_id: str, num_envs: int | None = None, seed: int = 42, cfg: "_abi.LtCfg | None" = None, object_sizes=None):
        """`cfg`: a complete lt_cfg (e.g. translated from the reference cfg tree) instead of the registration's preset."""
        self.cfg = cfg.copy() if cfg is not None else _abi.preset_cfg(task_id, num_envs=num_envs, seed=seed)
        num_envs = int(self.cfg.num_envs)
        self.o = oracle_lib.OracleEnv(self.cfg)
        if object_sizes is not None:
            self.cfg.obj_size_explicit = self.o.cfg.obj_size_explicit = 1
            Layout(num_envs, int(oracle_lib.load().lt_oracle_obs_dim(self.o.cfg)), int(self.cfg.tactile_enabled)).arr(self.o.arena, "LT_F_OBJ_SIZES")[:num_envs] = \
                np.asarray(object_sizes, dtype=np.float32)
        self.o.reset_all()
        self.num_envs, self.device = num_envs, torch.device("cpu")
        self.num_obs = int(oracle_lib.load().lt_oracle_obs_dim(self.o.cfg))
        self.num_privileged_obs = self.num_obs
        wide = int(self.cfg.tactile_format) in (_abi.CONSTS["LT_TACTILE_PROCESSED"], _abi.CONSTS["LT_TACTILE_ORIGINAL"])
        self.layout = Layout(num_envs, self.num_obs, int(self.cfg.tactile_enabled), 884 if wide else 442)
        self.max_episode_length = int(self.cfg.max_episode_length)
        self.step_dt = float(self.cfg.sim_dt) * int(self.cfg.decimation)

    def _arr(self, name: str) -> np.ndarray:
        return self.layout.arr(self.o.arena, name)[: self.num_envs]

    def field(self, name: str) -> torch.Tensor:
        """Quad field as [N, Q, 4] (component c = q*4 + lane), like LocoTouchVecEnv.field (a copy: the host arena is quad-major)."""
        if name in self.layout.plain:  # plain [N] / fixed-size fields: views of the host arena
            a = self.layout.arr(self.o.arena, name)
            return torch.from_numpy(a[: self.num_envs] if a.shape[0] >= self.num_envs and name not in ("LT_F_CMD_PARAMS", "LT_F_COUNTERS", "LT_F_GATE_RING") else a)
        v = self.layout.vec(self.o.arena, name)
        return torch.from_numpy(v).reshape(self.num_envs, -1, 4)

    @property
    def unwrapped(self):
        return self

    @property
    def episode_length_buf(self) -> torch.Tensor:
        return torch.from_numpy(self._arr("LT_F_EP_LEN"))

    @episode_length_buf.setter
    def episode_length_buf(self, value: torch.Tensor) -> None:
        self._arr("LT_F_EP_LEN")[:] = value.detach().cpu().numpy()

    def get_observations(self):
        obs = torch.from_numpy(self._arr("LT_F_OBS_POLICY"))
        groups = {"policy": obs, "critic": torch.from_numpy(self._arr("LT_F_OBS_CRITIC"))}
        if self.cfg.tactile_enabled:  # student tasks: same groups as LocoTouchVecEnv (object_state = the policy rows' object block)
            groups["tactile"] = torch.from_numpy(self._arr("LT_F_OBS_TACTILE"))
            if self.cfg.tactile_aux_groups & 1:
                groups["original_tactile"] = torch.from_numpy(self._arr("LT_F_OBS_TACTILE_ORIGINAL"))
            if self.cfg.tactile_aux_groups & 2:
                groups["processed_tactile"] = torch.from_numpy(self._arr("LT_F_OBS_TACTILE_PROCESSED"))
            groups["object_state"] = obs[:, self.num_obs - 13 * int(self.cfg.obs_history):]
        return obs, {"observations": groups}

    def reset(self):
        return self.get_observations()

    def curriculum_sync(self, dist, nsteps: int) -> None:
        """LocoTouchVecEnv.curriculum_sync on the host arena."""
        if not self.cfg.cur_gate_external:
            return
        ring = torch.from_numpy(self.layout.arr(self.o.arena, "LT_F_GATE_RING").copy())
        dist.all_reduce_sum_(ring)
        self.o.curriculum_apply_global(ring.numpy(), nsteps, self.num_envs * dist.world_size)

    def step(self, actions: torch.Tensor):
        self.o.step(actions.detach().cpu().numpy().astype(np.float32))
        obs, extras = self.get_observations()
        extras["time_outs"] = torch.from_numpy(self._arr("LT_F_TIME_OUT").astype(bool))
        extras["log"] = {}
        return obs, torch.from_numpy(self._arr("LT_F_REWARD")), torch.from_numpy(self._arr("LT_F_DONES")), extras

    def request_termination(self, mask: torch.Tensor, time_out: bool = False) -> None:
        """LocoTouchVecEnv.request_termination on the host arena."""
        from locotouch_amd import _abi

        bits = self._arr("LT_F_TERM_BITS")
        bits |= (mask.cpu().numpy().astype(np.int32) << _abi.CONSTS["LT_TIMEOUT_REQUEST_BIT" if time_out else "LT_TERM_REQUEST_BIT"])

    @property
    def cmd_params(self) -> torch.Tensor:
        return torch.from_numpy(self.layout.arr(self.o.arena, "LT_F_CMD_PARAMS"))

    def close(self):
        pass
