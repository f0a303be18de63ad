"""Host-side mirror of the reference's env interface on top of the HIP library (C ABI: include/lt_env.h).

`LocoTouchVecEnv` implements the `VecEnv` protocol that loco_rl's runner drives (reference
loco_rl/loco_rl/env/vec_env.py:12-101; call sites loco_rl/loco_rl/runners/on_policy_runner.py:32,44,81,121-132,158)
with the semantics of IsaacLab's `RslRlVecEnvWrapper` around `ManagerBasedRLEnv`:

    obs, extras = env.get_observations()
    obs, rew, dones, extras = env.step(actions)      # extras["observations"]["critic"], extras["time_outs"], extras["log"]
    env.episode_length_buf = torch.randint_like(...)  # settable (on_policy_runner.py:121-124)

PyTorch is plumbing only: it owns the device arena (one uint8 tensor) and hands `data_ptr()`/the current stream to
the C ABI; every tensor this class returns is a zero-copy view into that arena.  There is no CPU fallback: a missing
liblocotouch_env.so or a non-CUDA(HIP) device raises.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _abi

C = _abi.CONSTS
_TORCH_DTYPES = {0: torch.float32, 1: torch.int64, 2: torch.uint8, 3: torch.int32}

def task_ids() -> list[str]:
    """gym ids this build resolves (lt_cfg_preset): the reference registry of locotouch/config/locotouch/__init__.py."""
    return _abi.preset_ids()


REWARD_TERM_NAMES = [  # manager order == enum lt_reward_term
    "alive", "track_lin_vel_xy", "track_ang_vel_z", "foot_slip", "foot_dragging", "gait", "track_base_height",
    "base_z_velocity", "base_roll_pitch_angle", "base_roll_pitch_velocity", "joint_position_limit", "joint_position",
    "joint_acceleration", "joint_velocity", "joint_torque", "action_rate", "thigh_calf_collision", "object_xy_position",
    "object_xy_velocity", "object_z_contact", "object_z_velocity", "object_roll_pitch_angle", "object_roll_pitch_velocity",
    "object_yaw_alignment", "object_dangerous_state"]
TERMINATION_NAMES = ["time_out", "base_orientation", "base_height_below_minimum", "base_contact", "hip_contact",
                     "object_below_robot", "object_bad_orientation"]


class LocoTouchVecEnv:
    """One process per GPU; `num_envs` environments sharded to this rank."""

    def __init__(self, task: str | int = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs: int | None = None,
                 device: str | torch.device = "cuda:0", seed: int = 42, cfg: _abi.LtCfg | None = None,
                 object_sizes: torch.Tensor | None = None, **overrides):
        """`task`: a registered gym id (its preset; `num_envs=None` keeps the registration's default, e.g. 50 for -Play-) or
        an LT_TASK_* kind.  `cfg`: a complete lt_cfg instead (e.g. translated from a reference cfg tree,
        locotouch_amd/compat/cfg_translate.py).  `object_sizes` [N][2]: explicit per-env cylinder (radius, length)."""
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("LocoTouchVecEnv needs a HIP device (there is no CPU path in the product; "
                               "the CPU oracle under oracle/ is test infrastructure only)")
        self._lib = _abi.load()
        if cfg is not None:
            self.cfg = cfg.copy()
        elif isinstance(task, str):
            self.cfg = _abi.preset_cfg(task, num_envs=num_envs, seed=seed)
        else:
            self.cfg = _abi.default_cfg(int(task), num_envs=4096 if num_envs is None else num_envs, seed=seed)
        if object_sizes is not None:
            self.cfg.obj_size_explicit = 1
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(f"lt_cfg has no field {k!r}")
            setattr(self.cfg, k, v)
        self.task = task
        self.num_envs = int(self.cfg.num_envs)
        self.num_actions = 12
        self.num_obs = self._lib.lt_cfg_obs_dim(ctypes.byref(self.cfg))
        self.num_privileged_obs = self.num_obs
        self.step_dt = float(self.cfg.sim_dt) * int(self.cfg.decimation)
        self.max_episode_length_s = float(self.cfg.episode_length_s)
        self.max_episode_length = int(self.cfg.max_episode_length)
        self._handle = ctypes.c_void_p()
        _abi.check(self._lib.lt_env_create(ctypes.byref(self.cfg), ctypes.byref(self._handle)), "lt_env_create")
        nbytes = ctypes.c_size_t()
        _abi.check(self._lib.lt_env_state_bytes(ctypes.byref(self.cfg), ctypes.byref(nbytes)), "lt_env_state_bytes")
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(nbytes.value + 256, dtype=torch.uint8, device=self.device)
        pad = (-self.arena.data_ptr()) % 256
        self._arena_aligned = self.arena[pad:pad + nbytes.value]
        _abi.check(self._lib.lt_env_bind(self._handle, ctypes.c_void_p(self._arena_aligned.data_ptr()), nbytes.value), "lt_env_bind")
        self._views: dict[int, torch.Tensor] = {}
        self.obs_policy = self.view(C["LT_F_OBS_POLICY"])
        self.obs_critic = self.view(C["LT_F_OBS_CRITIC"])
        self.reward_buf = self.view(C["LT_F_REWARD"])
        self.dones_buf = self.view(C["LT_F_DONES"])
        self.terminated_buf = self.view(C["LT_F_TERMINATED"])
        self.time_out_buf = self.view(C["LT_F_TIME_OUT"])
        self._ep_len = self.view(C["LT_F_EP_LEN"])
        self.cmd_params = self.view(C["LT_F_CMD_PARAMS"])
        self.counters = self.view(C["LT_F_COUNTERS"])
        # student tasks (cfg.tactile_enabled): observation groups `tactile` [N][442] and `object_state` [N][78]
        # (reference object_transport_student_env_cfg.py:165-166; the distillation reads them, distillation.py:57-59,159-162)
        self.tactile = bool(self.cfg.tactile_enabled)
        self.obs_tactile = self.view(C["LT_F_OBS_TACTILE"]) if self.tactile else None
        # the student -Play- env's 4-channel groups [N][884] (object_transport_student_env_cfg.py:170-171; cfg.tactile_aux_groups)
        aux = int(self.cfg.tactile_aux_groups) if self.tactile else 0
        self.obs_tactile_original = self.view(C["LT_F_OBS_TACTILE_ORIGINAL"]) if aux & 1 else None
        self.obs_tactile_processed = self.view(C["LT_F_OBS_TACTILE_PROCESSED"]) if aux & 2 else None
        self.obs_object_state = self.view(C["LT_F_OBS_OBJECT_STATE"]) if self.cfg.task == C["LT_TASK_TRANSPORT_TEACHER"] else None
        self.extras: dict = {}
        self._log_finished = None
        if self.cfg.obj_size_explicit:
            if object_sizes is None or tuple(object_sizes.shape) != (self.num_envs, 2):
                raise ValueError("cfg.obj_size_explicit needs object_sizes [num_envs][2] (radius, length)")
            self.view(C["LT_F_OBJ_SIZES"]).copy_(object_sizes.to(device=self.device, dtype=torch.float32))
        self.reset()

    # ---- zero-copy views -------------------------------------------------------------------------
    def view(self, field: int) -> torch.Tensor:
        if field in self._views:
            return self._views[field]
        v = _abi.LtView()
        _abi.check(self._lib.lt_env_get_view(self._handle, field, ctypes.byref(v)), "lt_env_get_view")
        dtype = _TORCH_DTYPES[v.dtype]
        esz = torch.empty((), dtype=dtype).element_size()
        off = v.ptr - self._arena_aligned.data_ptr()
        assert off >= 0 and off % esz == 0
        base = self._arena_aligned.view(dtype) if esz == 1 else self._arena_aligned[: self._arena_aligned.numel() // esz * esz].view(dtype)
        shape = tuple(v.shape[i] for i in range(v.ndim))
        stride = tuple(v.stride[i] for i in range(v.ndim))
        t = torch.as_strided(base, shape, stride, storage_offset=off // esz)
        self._views[field] = t
        return t

    def field(self, name: str) -> torch.Tensor:
        """Quad fields come back as [N, Q, 4] views with component c = q*4 + lane (include/lt_layout.h)."""
        return self.view(C[name])

    # ---- VecEnv protocol -------------------------------------------------------------------------
    @property
    def episode_length_buf(self) -> torch.Tensor:
        return self._ep_len

    @episode_length_buf.setter
    def episode_length_buf(self, value: torch.Tensor) -> None:
        self._ep_len.copy_(value.to(self._ep_len.dtype))

    @property
    def unwrapped(self):
        return self

    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        _abi.check(self._lib.lt_env_reset_all(self._handle, self._stream()), "lt_env_reset_all")
        return self.get_observations()

    def _extras(self) -> dict:
        groups = {"policy": self.obs_policy, "critic": self.obs_critic}
        if self.tactile:
            groups["tactile"] = self.obs_tactile
            groups["object_state"] = self.obs_object_state
            if self.obs_tactile_original is not None:
                groups["original_tactile"] = self.obs_tactile_original
            if self.obs_tactile_processed is not None:
                groups["processed_tactile"] = self.obs_tactile_processed
        return {"observations": groups, "time_outs": self.time_out_buf.bool()}

    def tactile_update(self) -> None:
        """lt_env_tactile_update: for drivers of the launch-only row/rollout entry points (lt_env_step runs it itself)."""
        _abi.check(self._lib.lt_env_tactile_update(self._handle, self._stream()), "lt_env_tactile_update")

    def get_observations(self):
        return self.obs_policy, self._extras()

    def step(self, actions: torch.Tensor):
        """One env step for all envs; returns views that the next step overwrites (same ownership rule as the
        reference: the trainer copies what it keeps, loco_rl/loco_rl/storage/rollout_storage.py:86-97)."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.shape != (self.num_envs, 12):
            actions = actions.to(torch.float32).reshape(self.num_envs, 12).contiguous()
        _abi.check(self._lib.lt_env_step(self._handle, ctypes.c_void_p(actions.data_ptr()), self._stream()), "lt_env_step")
        return self.obs_policy, self.reward_buf, self.dones_buf, self._extras()

    def step_raw(self, actions_ptr: int) -> None:
        """Launch-only variant for captured (hipGraph) rollouts: no tensor bookkeeping on the host."""
        _abi.check(self._lib.lt_env_step(self._handle, ctypes.c_void_p(actions_ptr), self._stream()), "lt_env_step")

    def step_rows_raw(self, actions_ptr: int, prev_policy: int, prev_critic: int, next_policy: int, next_critic: int) -> None:
        """lt_env_step_rows: observation rows read from / written to caller storage (0 = arena rows)."""
        vp = ctypes.c_void_p
        _abi.check(self._lib.lt_env_step_rows(self._handle, vp(actions_ptr), vp(prev_policy or None), vp(prev_critic or None),
                                              vp(next_policy or None), vp(next_critic or None), self._stream()), "lt_env_step_rows")

    def step_rollout_raw(self, actions_ptr: int, prev_policy: int, prev_critic: int, next_policy: int, next_critic: int, values_ptr: int,
                         gamma: float, st_rewards_ptr: int, st_dones_ptr: int) -> None:
        """lt_env_step_rollout: step_rows_raw + the storage writes of the transition (bootstrapped reward, dones)."""
        vp = ctypes.c_void_p
        _abi.check(self._lib.lt_env_step_rollout(self._handle, vp(actions_ptr), vp(prev_policy or None), vp(prev_critic or None),
                                                 vp(next_policy or None), vp(next_critic or None), vp(values_ptr), float(gamma),
                                                 vp(st_rewards_ptr), vp(st_dones_ptr), self._stream()), "lt_env_step_rollout")

    def defer_gate(self, mode: int) -> None:
        """lt_env_defer_gate (include/lt_env.h): where the population pass of a step (curriculum decision, population gate, step
        counter) runs - 0 behind every step (default), 1 the caller's `gate_update()`, 2 chained into the next step launch (a
        chain ends with `gate_update()`; until then the command block and the counters lag)."""
        _abi.check(self._lib.lt_env_defer_gate(self._handle, int(mode)), "lt_env_defer_gate")

    def gate_update(self) -> None:
        """The outstanding population pass, if any (lt_env_gate_update)."""
        _abi.check(self._lib.lt_env_gate_update(self._handle, self._stream()), "lt_env_gate_update")

    def set_row_format(self, dtype) -> None:
        """Element format of the rows behind step_rows_raw / step_rollout_raw pointers: torch.float32 or torch.bfloat16
        (lt_env_set_row_format; the arena's own rows stay f32)."""
        import torch

        fmt = {torch.float32: _abi.CONSTS["LT_ROWS_F32"], torch.bfloat16: _abi.CONSTS["LT_ROWS_BF16"]}[dtype]
        _abi.check(self._lib.lt_env_set_row_format(self._handle, fmt), "lt_env_set_row_format")

    def check(self) -> None:
        """Raise if a chained step launch lost its population-pass announcement (lt_env_check; waits for the stream)."""
        _abi.check(self._lib.lt_env_check(self._handle, self._stream()), "lt_env_check")

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._handle

    def step_profiled(self, actions: torch.Tensor) -> float:
        """lt_env_step with HIP events around the step kernel; returns its duration in ms (host-syncing)."""
        ms = ctypes.c_float()
        _abi.check(self._lib.lt_env_step_profiled(self._handle, ctypes.c_void_p(actions.data_ptr()), self._stream(), ctypes.byref(ms)),
                   "lt_env_step_profiled")
        return float(ms.value)

    def step_rows_profiled(self, actions: torch.Tensor, prev_policy: torch.Tensor, prev_critic: torch.Tensor, next_policy: torch.Tensor,
                           next_critic: torch.Tensor) -> float:
        """lt_env_step_rows with HIP events around the step kernel (rows in the current row format); duration in ms."""
        ms = ctypes.c_float()
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        _abi.check(self._lib.lt_env_step_rows_profiled(self._handle, p(actions), p(prev_policy), p(prev_critic), p(next_policy),
                                                       p(next_critic), self._stream(), ctypes.byref(ms)), "lt_env_step_rows_profiled")
        return float(ms.value)

    def request_termination(self, mask: torch.Tensor, time_out: bool = False) -> None:
        """End the envs of `mask` (bool [N]) in the NEXT step: sets LT_TERM_REQUEST_BIT (or, `time_out`, LT_TIMEOUT_REQUEST_BIT) in
        their LT_F_TERM_BITS word (include/lt_env.h, LT_T_USER / LT_T_USER_TIME_OUT - the hook of termination terms outside the
        fused set, compat/scene_views.py)."""
        bits = self.view(C["LT_F_TERM_BITS"])
        bits.bitwise_or_(mask.to(device=self.device, dtype=torch.int32) << C["LT_TIMEOUT_REQUEST_BIT" if time_out else "LT_TERM_REQUEST_BIT"])

    def eval_terms(self) -> None:
        _abi.check(self._lib.lt_env_eval_terms(self._handle, self._stream()), "lt_env_eval_terms")

    def curriculum_update(self, records: torch.Tensor) -> None:
        """The curriculum pass of the step kernel's tail on caller-supplied records [N][4] (parity-test hook)."""
        assert records.dtype == torch.float32 and records.is_contiguous() and records.shape == (self.num_envs, 4) and records.is_cuda
        _abi.check(self._lib.lt_env_curriculum_update(self._handle, ctypes.c_void_p(records.data_ptr()), self._stream()),
                   "lt_env_curriculum_update")

    def curriculum_sync(self, dist, nsteps: int) -> None:
        """Multi-rank curriculum gate (cfg.cur_gate_external): all-reduce the population sums of the last
        `nsteps` curriculum passes over the ranks and replay the reference's decision sequence on them - every rank widens the
        command ranges on the same step.  One 1-KiB all-reduce per rollout; no host sync."""
        if not self.cfg.cur_gate_external:
            return
        ring = self.view(C["LT_F_GATE_RING"]).clone()
        dist.all_reduce_sum_(ring)
        _abi.check(self._lib.lt_env_curriculum_apply_global(self._handle, ctypes.c_void_p(ring.data_ptr()), int(nsteps),
                                                            int(self.num_envs) * int(dist.world_size), self._stream()),
                   "lt_env_curriculum_apply_global")
        self._gate_ring_keepalive = ring  # the kernel reads it asynchronously

    def set_command_ranges(self, ranges, zero_steps: int, rel_standing: float) -> None:
        arr = (ctypes.c_float * 6)(*[float(x) for x in ranges])
        _abi.check(self._lib.lt_env_set_command_ranges(self._handle, arr, int(zero_steps), float(rel_standing), self._stream()),
                   "lt_env_set_command_ranges")

    # ---- logging (host sync only when asked) -------------------------------------------------------
    def episode_log(self) -> dict:
        """`extras["log"]`-style means over the episodes finished since the last call
        (Episode_Reward/<term> = mean(sum)/max_episode_length_s, RewardManager.reset [DEP])."""
        # the log is read once per training iteration and waits for the stream anyway: the place to turn a lost chained hand-off
        # (LT_F_COUNTERS[1], lt_env_defer_gate mode 2) into an error instead of training on a stale command block
        self.check()
        info = self.field("LT_F_LAST_EPISODE_INFO")[:, 0, :]
        finished = info[:, 0].clone()
        prev = self._log_finished if self._log_finished is not None else torch.zeros_like(finished)
        mask = finished > prev
        self._log_finished = finished
        log = {}
        if bool(mask.any()):
            sums = self.field("LT_F_LAST_EPISODE_SUMS").reshape(self.num_envs, -1)[mask]
            for i, name in enumerate(REWARD_TERM_NAMES):
                if self.cfg.reward_weight[i] != 0:
                    log[f"Episode_Reward/{name}"] = float(sums[:, i].mean()) / self.max_episode_length_s
            bits = info[mask, 2].to(torch.int64)
            for b, name in enumerate(TERMINATION_NAMES):
                if self.cfg.term_enabled[b]:
                    log[f"Episode_Termination/{name}"] = float(((bits >> b) & 1).float().sum())
            log["Episode/length"] = float(info[mask, 1].mean())
            log["Episode/reward"] = float(sums[:, : len(REWARD_TERM_NAMES)].sum(1).mean())
        log.update(self.command_metrics(mask))
        p = self.cmd_params
        log["Metrics/base_velocity/lin_vel_x"] = float(p[1])
        log["Metrics/base_velocity/lin_vel_y"] = float(p[3])
        log["Metrics/base_velocity/ang_vel_z"] = float(p[5])
        log["Metrics/base_velocity/initial_zero_command_steps"] = float(p[15])
        log["Metrics/base_velocity/rel_standing_envs"] = float(p[16])
        return log

    def command_metrics(self, mask: torch.Tensor | None = None) -> dict:
        """The command term's `Metrics/base_velocity/*` (reference mdp/commands.py:392-417; IsaacLab CommandTerm.reset [DEP] logs the
        MEAN OVER THE ENVS THAT RESET IN A STEP of the values the last `compute()` left, and the runner averages those per-step
        entries over the iteration, loco_rl/loco_rl/runners/on_policy_runner.py log()).

        error_vel_xy, error_vel_yaw, foot_air_time_variance: the step kernel keeps each env's values as of the end of the last step
        and snapshots them, with the step id, in the env's reset path (LT_F_LAST_CMD_METRICS); here the envs in `mask` (those that
        finished an episode since the last log read; None: every env that has finished one) are grouped by step id -> one mean per
        reset batch -> the mean of the batches, the runner's number.  An env that reset twice since the last read counts with its
        later reset only.
        The six gait statistics (foot_step_frequency ...) are POPULATION means in the reference (broadcast to every env by
        `metrics[...][:] = average`, commands.py:404-417): they are formed here from the current state of all envs - the reference
        logs the same quantity as of the step before each reset batch."""
        f = self.field
        out = {}
        last = f("LT_F_LAST_CMD_METRICS")[:, 0, :]
        if mask is None:
            mask = self.field("LT_F_LAST_EPISODE_INFO")[:, 0, 0] > 0
        if bool(mask.any()):
            for name, v in zip(("error_vel_xy", "error_vel_yaw", "foot_air_time_variance"), reset_batch_means(last[mask])):
                out[f"Metrics/base_velocity/{name}"] = v
        valid = f("LT_F_GAIT_VALID_LAST_AIR")[:, 0, :][:, [0, 3, 1, 2]]  # gait-class column order [FR, RL, FL, RR]
        ok = (valid > 1.0e-6).all(dim=1)
        stats = {"foot_step_frequency": 0.0, "pair_1_step_frequency": 0.0, "pair_2_step_frequency": 0.0,
                 "step_air_time": 0.0, "pair_1_air_time": 0.0, "pair_2_air_time": 0.0}
        if bool(ok.any()):
            m = valid[ok]
            for name, cols in (("", slice(None)), ("pair_1_", [0, 1]), ("pair_2_", [2, 3])):
                t = float(m[:, cols].mean())
                stats[("foot_step_frequency" if not name else name + "step_frequency")] = (0.5 / t) if t > 0 else 0.0
                stats[("step_air_time" if not name else name + "air_time")] = t if t > 0 else 0.0
        out.update({f"Metrics/base_velocity/{k}": v_ for k, v_ in stats.items()})
        return out

    def current_command_metrics(self) -> dict:
        """The per-env metric tensors as `command_manager.get_term("base_velocity").metrics` holds them between two steps."""
        t = self.field("LT_F_EVENT_TIMERS")[:, 0, :]
        return {"error_vel_xy": t[:, 2], "error_vel_yaw": t[:, 3], "foot_air_time_variance": self.field("LT_F_TRUNK_FORCE_HIST")[:, 0, 3]}

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.lt_env_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reset_batch_means(rows: torch.Tensor) -> list[float]:
    """rows [k][4] = LT_F_LAST_CMD_METRICS of k envs (three metric values + the step id of the env's last reset): the mean over the
    reset batches (one per distinct step id) of each batch's mean - what the reference's runner reports for a window of steps
    (CommandTerm.reset logs a batch mean per step, the runner averages the steps)."""
    ids, inv = torch.unique(rows[:, 3], return_inverse=True)
    cnt = torch.zeros(ids.numel(), device=rows.device, dtype=rows.dtype).index_add_(0, inv, torch.ones_like(rows[:, 0]))
    return [float((torch.zeros_like(cnt).index_add_(0, inv, rows[:, i]) / cnt).mean()) for i in range(3)]


def make(task: str, num_envs: int | None = None, device: str = "cuda:0", seed: int = 42, **kw) -> LocoTouchVecEnv:
    """`gym.make(task, cfg=...)` + `RslRlVecEnvWrapper(env)` equivalent (reference locotouch/scripts/train.py:98,116)."""
    return LocoTouchVecEnv(task, num_envs=num_envs, device=device, seed=seed, **kw)


__all__ = ["LocoTouchVecEnv", "make", "task_ids", "reset_batch_means", "REWARD_TERM_NAMES", "TERMINATION_NAMES"]
_ = math
