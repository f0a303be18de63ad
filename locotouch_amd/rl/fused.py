"""Fused rollout step on the GPU: policy GEMMs (PyTorch-ROCm / hipBLASLt) -> lt_rollout_act -> lt_env_step -> lt_rollout_record.

Semantically one iteration of the reference's rollout loop (loco_rl/loco_rl/runners/on_policy_runner.py:154-199):
`actions = alg.act(obs, critic_obs); obs, rew, dones, infos = env.step(actions); alg.process_env_step(rew, dones, infos)`,
with the ~30 elementwise launches between the GEMMs and the env step replaced by the two fused kernels of
locotouch_amd/csrc/lt_rollout.hip, the critic MLP running on a forked stream beside the actor MLP, and no host sync, so a
whole 24-step rollout captures into one hipGraph.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _abi


class FusedRollout:
    def __init__(self, env, alg):
        from ..env import LocoTouchVecEnv

        if not isinstance(env, LocoTouchVecEnv):
            raise TypeError("FusedRollout drives a LocoTouchVecEnv (HIP env)")
        ac = alg.actor_critic
        if getattr(ac, "noise_std_type", "scalar") != "scalar":
            raise ValueError("FusedRollout supports the 'scalar' noise_std_type of the LocoTouch agent configs")
        self.env, self.alg, self.lib = env, alg, _abi.load()
        self.device = env.device
        self.actions = torch.zeros(env.num_envs, 12, device=self.device)
        self.side = torch.cuda.Stream(device=self.device)
        self._step_counter = env.counters  # int64[4] view; [0] = common step counter (device-resident RNG key)

    @staticmethod
    def _p(t: torch.Tensor) -> ctypes.c_void_p:
        return ctypes.c_void_p(t.data_ptr())

    def step(self, t: int) -> None:
        env, alg, st, p = self.env, self.alg, self.alg.storage, self._p
        ac = alg.actor_critic
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            value = ac.critic(env.obs_critic)
        mu = ac.actor(env.obs_policy)
        main.wait_stream(self.side)
        value.record_stream(main)
        stream = ctypes.c_void_p(main.cuda_stream)
        n = env.num_envs
        _abi.check(self.lib.lt_rollout_act(n, env.num_obs, int(env.cfg.seed), p(self._step_counter), p(mu), p(ac.std.data), p(value),
                                           p(env.obs_policy), p(env.obs_critic), p(st.observations[t]), p(st.privileged_observations[t]),
                                           p(st.actions[t]), p(st.mu[t]), p(st.sigma[t]), p(st.values[t]), p(st.actions_log_prob[t]),
                                           p(self.actions), stream), "lt_rollout_act")
        env.step_raw(self.actions.data_ptr())
        _abi.check(self.lib.lt_rollout_record(n, float(alg.gamma), p(env.reward_buf), p(env.dones_buf), p(env.time_out_buf), p(st.values[t]),
                                              p(st.rewards[t]), p(st.dones[t]), stream), "lt_rollout_record")
        st.step = t + 1

    def rollout(self, num_steps: int) -> None:
        """`num_steps` consecutive steps into storage slots 0.. (inference mode, capturable)."""
        self.alg.storage.clear()
        with torch.inference_mode():
            for t in range(num_steps):
                self.step(t)
