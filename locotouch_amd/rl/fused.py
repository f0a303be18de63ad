"""Fused rollout step on the GPU, semantically one iteration of the reference's rollout loop
(loco_rl/loco_rl/runners/on_policy_runner.py:154-199):
`actions = alg.act(obs, critic_obs); obs, rew, dones, infos = env.step(actions); alg.process_env_step(rew, dones, infos)`.

Packed path (network shapes lt_mlp.hip covers - every LocoTouch agent config): launches on ONE stream,
    [lt_rollout_policy_value: actor + critic MLPs, sampling, log-prob, storage writes of actions/mu/sigma/values/log-prob]
 -> [lt_env_step_rollout: env step, observation rows written into storage slot t+1, bootstrapped reward + dones into slot t;
     the one-wave population pass of the PREVIOUS step (curriculum decision / population gate) rides in this launch, on an idle
     wave beside the physics (lt_env_defer_gate mode 2); one explicit pass closes the rollout]
with no host sync, so a whole 24-step rollout captures into one hipGraph.  The chain is kept linear on purpose: forked
streams turn graph edges into cross-queue dependencies that cost more than the overlap returns (measured again with the
population pass on a side branch beside the next policy launch, lt_env_defer_gate: 99.8 us per step against 86.3 us linear).

Torch path (shapes outside lt_mlp's limits): torch GEMMs -> lt_rollout_act -> lt_env_step_rows -> lt_rollout_record, with the
critic on a side stream.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _abi


class FusedRollout:
    """Drives a LocoTouchVecEnv and a PPO instance through rollout steps without leaving the device (module docstring).

    The env writes observation rows straight into the storage slots (slot t+1 from slot t; the last step writes the arena
    rows), so nothing copies observations."""

    def __init__(self, env, alg, use_packed_mlp: bool = True):
        from ..env import LocoTouchVecEnv

        if not isinstance(env, LocoTouchVecEnv):
            raise TypeError("FusedRollout drives a LocoTouchVecEnv (HIP env)")
        ac = alg.actor_critic
        if getattr(ac, "noise_std_type", "scalar") != "scalar":
            raise ValueError("FusedRollout supports the 'scalar' noise_std_type of the LocoTouch agent configs")
        self.env, self.alg, self.lib = env, alg, _abi.load()
        self.device = env.device
        self.actions = torch.zeros(env.num_envs, 12, device=self.device)
        # Philox key of the policy noise: the kernels key the draws by the LOCAL row index, so rank r (env_index_offset = r * N) gets its
        # own key - with the population's seed alone every rank would explore with the same noise on its env i (offset 0: the seed itself)
        self._noise_seed = (int(env.cfg.seed) + 0x9E3779B97F4A7C15 * int(env.cfg.env_index_offset)) & 0xFFFFFFFFFFFFFFFF
        self.side = torch.cuda.Stream(device=self.device)
        # policy-noise RNG key: a copy of the env's common step counter (refreshed at every rollout start, advanced by
        # lt_rollout_record) so that lt_rollout_act never has to wait for the overlapped post kernel
        self._act_counter = env.counters[:1].clone()
        # rows written in place into the storage need whole 16-env wave tiles (npad == n)
        self.rows_in_storage = env.num_envs % 16 == 0
        # single-launch MLPs (lt_mlp.hip) when the stacks fit its shape limits, else the torch modules
        self.actor_mlp = self.critic_mlp = None
        if use_packed_mlp:
            from .mlp import PackedMLP, describe

            if describe(ac.actor) is not None and describe(ac.critic) is not None and env.num_actions == 12:
                self.actor_mlp, self.critic_mlp = PackedMLP(ac.actor), PackedMLP(ac.critic)
        # bf16 observation rows (BASELINE config 5): the storage slots hold bf16 rows; the env kernel reads slot t and writes slot
        # t + 1 as bf16 (the newest frame rounded once, older frames carried bit for bit), the policy kernel widens them in LDS.
        # The rows behind the LAST step go to two extra bf16 buffers and from there to the arena's f32 rows (exact).
        self.obs_dtype = alg.storage.observations.dtype
        self._tail_rows = None
        if self.obs_dtype == torch.bfloat16:
            if self.actor_mlp is None or not self.rows_in_storage:
                raise ValueError("bf16 observation rows need the packed MLP path and num_envs % 16 == 0")
            self.actor_mlp.set_input_format(torch.bfloat16)
            self.critic_mlp.set_input_format(torch.bfloat16)
            st = alg.storage
            self._tail_rows = (torch.zeros_like(st.observations[0]), torch.zeros_like(st.privileged_observations[0]))

    @staticmethod
    def _p(t: torch.Tensor) -> ctypes.c_void_p:
        return ctypes.c_void_p(t.data_ptr())

    def step(self, t: int, last: bool) -> None:
        if self.actor_mlp is not None:
            self._step_packed(t, last)
        else:
            self._step_torch(t, last)
        self.alg.storage.step = t + 1

    def _rows(self, t: int, last: bool):
        """(obs, critic obs, next obs ptr, next critic obs ptr) of step t."""
        env, st = self.env, self.alg.storage
        if not self.rows_in_storage:
            return env.obs_policy, env.obs_critic, 0, 0
        if last and self._tail_rows is not None:
            nxt = (self._tail_rows[0].data_ptr(), self._tail_rows[1].data_ptr())
        else:
            nxt = (0, 0) if last else (st.observations[t + 1].data_ptr(), st.privileged_observations[t + 1].data_ptr())
        return st.observations[t], st.privileged_observations[t], nxt[0], nxt[1]

    @property
    def launches_per_step(self) -> int:
        """Kernel launches of one rollout step (the reference-shaped eager loop needs ~30)."""
        return 2 if self.actor_mlp is not None else 11  # policy + value, env step (the population pass rides in the next step's launch)

    def policy_value_launch(self, t: int) -> None:
        """The MLP launch of step t alone (bench.py times it for the MFMA roofline entry)."""
        env, alg, st, p = self.env, self.alg, self.alg.storage, self._p
        obs, cobs, _, _ = self._rows(t, False)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _abi.check(self.lib.lt_rollout_policy_value(
            ctypes.byref(self.actor_mlp.desc), p(self.actor_mlp.packed), p(obs), ctypes.byref(self.critic_mlp.desc),
            p(self.critic_mlp.packed), p(cobs), p(st.values[t]), env.num_envs, self._noise_seed, p(self._act_counter), t,
            p(alg.actor_critic.std.data), p(st.actions[t]), p(st.mu[t]), p(st.sigma[t]), p(st.actions_log_prob[t]), p(self.actions), stream),
            "lt_rollout_policy_value")

    def _step_packed(self, t: int, last: bool) -> None:
        """Two launches on ONE stream (a linear graph: cross-queue edges of a forked graph cost more than they return on this stack):
        [actor + critic MLPs + sampling] -> [env step + storage record (+ the previous step's population pass on an idle wave)]."""
        env, alg, st, p = self.env, self.alg, self.alg.storage, self._p
        ac = alg.actor_critic
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        n = env.num_envs
        obs, cobs, nxt_p, nxt_c = self._rows(t, last)
        if not self.rows_in_storage:
            st.observations[t].copy_(obs)
            st.privileged_observations[t].copy_(cobs)
        _abi.check(self.lib.lt_rollout_policy_value(
            ctypes.byref(self.actor_mlp.desc), p(self.actor_mlp.packed), p(obs), ctypes.byref(self.critic_mlp.desc),
            p(self.critic_mlp.packed), p(cobs), p(st.values[t]), n, self._noise_seed, p(self._act_counter), t, p(ac.std.data),
            p(st.actions[t]), p(st.mu[t]), p(st.sigma[t]), p(st.actions_log_prob[t]), p(self.actions), stream),
            "lt_rollout_policy_value")
        prev = (obs.data_ptr(), cobs.data_ptr()) if self.rows_in_storage else (0, 0)
        env.step_rollout_raw(self.actions.data_ptr(), prev[0], prev[1], nxt_p, nxt_c, st.values[t].data_ptr(), float(alg.gamma),
                             st.rewards[t].data_ptr(), st.dones[t].data_ptr())

    def _step_torch(self, t: int, last: bool) -> None:
        """torch modules for the networks (shapes outside lt_mlp's limits): GEMMs -> lt_rollout_act -> env step -> record,
        critic and post pass on forked streams."""
        env, alg, st, p = self.env, self.alg, self.alg.storage, self._p
        ac = alg.actor_critic
        main = torch.cuda.current_stream(self.device)
        stream = ctypes.c_void_p(main.cuda_stream)
        n = env.num_envs
        null = ctypes.c_void_p(None)
        obs, cobs, nxt_p, nxt_c = self._rows(t, last)
        mu = ac.actor(obs)
        rows = (null, null, null, null) if self.rows_in_storage else (p(obs), p(cobs), p(st.observations[t]), p(st.privileged_observations[t]))
        _abi.check(self.lib.lt_rollout_act(n, env.num_obs, self._noise_seed, p(self._act_counter), p(mu), p(ac.std.data), null,
                                           *rows, p(st.actions[t]), p(st.mu[t]), p(st.sigma[t]), null,
                                           p(st.actions_log_prob[t]), p(self.actions), stream), "lt_rollout_act")
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            value = ac.critic(cobs)
        prev = (obs.data_ptr(), cobs.data_ptr()) if self.rows_in_storage else (0, 0)
        env.step_rows_raw(self.actions.data_ptr(), prev[0], prev[1], nxt_p, nxt_c)
        main.wait_stream(self.side)
        value.record_stream(main)
        _abi.check(self.lib.lt_rollout_record(n, float(alg.gamma), p(env.reward_buf), p(env.dones_buf), p(env.time_out_buf), p(value),
                                              p(st.rewards[t]), p(st.dones[t]), p(st.values[t]), p(self._act_counter), stream),
                   "lt_rollout_record")

    def rollout(self, num_steps: int) -> None:
        """`num_steps` consecutive steps into storage slots 0.. (inference mode, capturable)."""
        env, st = self.env, self.alg.storage
        st.clear()
        main = torch.cuda.current_stream(self.device)
        with torch.inference_mode():
            self._act_counter.copy_(env.counters[:1])
            if self.actor_mlp is not None:  # the optimizer has stepped since the last rollout
                self.actor_mlp.pack()
                self.critic_mlp.pack()
            if self.rows_in_storage:
                st.observations[0].copy_(env.obs_policy)
                st.privileged_observations[0].copy_(env.obs_critic)
            # chained steps (lt_env_defer_gate mode 2): the one-wave population pass of step t runs inside the launch of step t + 1,
            # beside its physics, so a rollout step is two launches; the pass of the last step closes the chain
            chain = not env.tactile  # (the tactile kernel keys its draws by the step counter, which lags inside a chain)
            if chain:
                env.defer_gate(2)
            if self._tail_rows is not None:
                env.set_row_format(torch.bfloat16)
            try:
                for t in range(num_steps):
                    self.step(t, last=t == num_steps - 1)
            finally:
                if chain:
                    env.gate_update()
                    env.defer_gate(0)
                if self._tail_rows is not None:
                    env.set_row_format(torch.float32)
                    env.obs_policy.copy_(self._tail_rows[0])
                    env.obs_critic.copy_(self._tail_rows[1])
