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

TASK_IDS = {"Isaac-Locomotion-LocoTouch-v1": "LT_TASK_LOCOMOTION", "Isaac-RandCylinderTransportTeacher-LocoTouch-v1": "LT_TASK_TRANSPORT_TEACHER"}


class OracleVecEnv:
    num_actions = 12

    def __init__(self, task_id: str, num_envs: int, seed: int = 42):
        self.cfg = _abi.default_cfg(_abi.CONSTS[TASK_IDS[task_id]], num_envs=num_envs, seed=seed)
        self.o = oracle_lib.OracleEnv(self.cfg)
        self.o.reset_all()
        self.num_envs, self.device = num_envs, torch.device("cpu")
        self.num_obs = int(oracle_lib.load().lt_oracle_obs_dim(self.o.cfg))
        self.num_privileged_obs = self.num_obs
        self.layout = Layout(num_envs, self.num_obs)
        self.max_episode_length = int(self.cfg.max_episode_length)
        self.step_dt = float(self.cfg.sim_dt) * int(self.cfg.decimation)

    def _arr(self, name: str) -> np.ndarray:
        return self.layout.arr(self.o.arena, name)[: self.num_envs]

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
        return obs, {"observations": {"policy": obs, "critic": torch.from_numpy(self._arr("LT_F_OBS_CRITIC"))}}

    def reset(self):
        return self.get_observations()

    def step(self, actions: torch.Tensor):
        self.o.step(actions.detach().cpu().numpy().astype(np.float32))
        obs, extras = self.get_observations()
        extras["time_outs"] = torch.from_numpy(self._arr("LT_F_TIME_OUT").astype(bool))
        extras["log"] = {}
        return obs, torch.from_numpy(self._arr("LT_F_REWARD")), torch.from_numpy(self._arr("LT_F_DONES")), extras

    def close(self):
        pass
