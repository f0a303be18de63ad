"""IsaacLab-layout views of the env state for manager terms the fused kernels do not know.

The reference's terms read `env.scene["robot"].data.joint_pos`, `env.scene.sensors["robot_contact_senosr"].data.net_forces_w_history`
(sic: the misspelt sensor name is part of the reference's API, config/base/locomotion_base_env_cfg.py:35),
`asset.data.body_lin_vel_w[:, body_ids]`, `env.action_manager.get_term("joint_pos").raw_actions`,
`env.reward_manager.get_term_cfg(name)`, `env.reward_manager._episode_sums[name]`, `env.termination_manager.terminated`,
`env.command_manager.get_command("base_velocity")`, ... (locotouch/mdp/rewards.py:31-56,423-466,569-604; commands.py:399-401;
curriculums.py:197-200,238,259).  The HIP env keeps its state in quad arrays (include/lt_layout.h); `TermEnv` presents them under
the reference's names and layouts - torch tensors built from the arena views on every access, on the env's device - so that the
reference's OWN term functions run unmodified as user terms (tests/test_reference_terms_slow_path.py does exactly that and
compares each with the fused kernel's column).  `ExtraTerms` evaluates user reward / termination terms on them after a step and
adds `weight * dt * value` to the reward the kernel computed (the SLOW path: a handful of torch launches per term and step).

Limits, stated:
* the step kernel resets finished envs inside the step, so a user term sees the post-reset state of an env that just finished and
  its value is dropped there (the fused terms are evaluated before the reset, as the RewardManager does [DEP]);
* contact forces are stored as norms - `net_forces_w_history` carries each body's |F| in the z component (norm-exact: every
  reference term takes `torch.linalg.norm(..., dim=-1)` of it; direction not kept); air / contact timers exist for the four
  feet (the reference's sensor has them for all 17 bodies; only the feet are read, rewards.py:79-92, commands.py:396);
* link poses / velocities other than the feet (`body_pos_w`, `body_quat_w`, `body_lin_vel_w`, `body_ang_vel_w`) are forward
  kinematics in torch from the root and joint state (the feet are the kernel's own LT_F_FOOT_POS_W / _VEL_W);
* user TERMINATION terms are evaluated on the state a step left and take effect through the kernel's own termination stage one
  env step LATER (LT_T_USER / LT_T_USER_TIME_OUT in include/lt_env.h: the reset has to happen inside the step kernel).
"""
from __future__ import annotations

import re

import torch

from .. import _abi
from . import math as M

BODY_NAMES = ["trunk"] + [f"{leg}_{part}" for part in ("hip", "thigh", "calf", "foot") for leg in ("a_FR", "b_FL", "c_RR", "d_RL")]
JOINT_NAMES = [f"{leg}_{part}_joint" for part in ("hip", "thigh", "calf") for leg in ("a_FR", "b_FL", "c_RR", "d_RL")]
ROBOT_SENSOR = "robot_contact_senosr"  # (sic) locomotion_base_env_cfg.py:35
# joint-frame origins in the parent link frame, leg order FR FL RR RL (include/lt_go1_model.h LT_JOINT_OFFSET_INIT / LT_FOOT_OFFSET_INIT;
# tests/test_reference_terms_slow_path.py checks these against the generated header)
HIP_OFFSET = [[0.1881, -0.04675, 0.0], [0.1881, 0.04675, 0.0], [-0.1881, -0.04675, 0.0], [-0.1881, 0.04675, 0.0]]
THIGH_OFFSET = [[0.0, -0.08, 0.0], [0.0, 0.08, 0.0], [0.0, -0.08, 0.0], [0.0, 0.08, 0.0]]
CALF_OFFSET = [0.0, 0.0, -0.213]
FOOT_OFFSET = [0.0, 0.0, -0.213]
REWARD_TERM_NAMES = [  # manager order == enum lt_reward_term (locotouch_amd/env.py)
    "alive", "track_lin_vel_xy", "track_ang_vel_z", "foot_slip", "foot_dragging", "gait", "track_base_height",
    "base_z_velocity", "base_roll_pitch_angle", "base_roll_pitch_velocity", "joint_position_limit", "joint_position",
    "joint_acceleration", "joint_velocity", "joint_torque", "action_rate", "thigh_calf_collision", "object_xy_position",
    "object_xy_velocity", "object_z_contact", "object_z_velocity", "object_roll_pitch_angle", "object_roll_pitch_velocity",
    "object_yaw_alignment", "object_dangerous_state"]
TERMINATION_NAMES = ["time_out", "base_orientation", "base_height_below_minimum", "base_contact", "hip_contact",
                     "object_below_robot", "object_bad_orientation"]
GAIT_FOOT_ORDER = [0, 3, 1, 2]  # the gait class's column order [FR, RL, FL, RR] = pair 0 then pair 1 (rewards.py:89-92)


def _find(keys, names, preserve_order=False):
    keys = [keys] if isinstance(keys, str) else list(keys)
    if preserve_order:
        ids = [i for k in keys for i, n in enumerate(names) if re.fullmatch(k, n)]
    else:
        ids = [i for i, n in enumerate(names) if any(re.fullmatch(k, n) for k in keys)]
    return ids, [names[i] for i in ids]


class _Data:
    """Attribute access -> a freshly built tensor (the quad arrays change every step; nothing is cached)."""

    def __init__(self, getters: dict):
        self._g = getters

    def __getattr__(self, name):
        try:
            return self._g[name]()
        except KeyError:
            raise AttributeError(f"no view for data.{name}; available: {sorted(self._g)}") from None


class _Entity:
    def __init__(self, data: _Data, body_names=(), joint_names=()):
        self.data, self.body_names, self.joint_names = data, list(body_names), list(joint_names)
        self.num_bodies, self.num_joints = len(self.body_names), len(self.joint_names)

    def find_bodies(self, keys, preserve_order=False):
        return _find(keys, self.body_names, preserve_order)

    def find_joints(self, keys, preserve_order=False):
        return _find(keys, self.joint_names, preserve_order)


class Scene:
    def __init__(self, entities: dict, sensors: dict, num_envs: int):
        self._e, self.sensors, self.num_envs = entities, sensors, num_envs

    def __getitem__(self, name):
        return self._e[name] if name in self._e else self.sensors[name]

    def keys(self):
        return list(self._e) + list(self.sensors)


class _CommandTermView:
    """`env.command_manager.get_term("base_velocity")`: the attributes terms reach for (commands.py:379-576; curriculums.py:187-193)."""

    def __init__(self, env):
        self._env = env

    @property
    def vel_command_b(self):
        frozen = self._env.cmd_frozen  # (inside ExtraTerms' evaluation: the command the step's reward stage saw)
        return frozen.clone() if frozen is not None else self._env.vec.field("LT_F_CMD")[:, 0, :3].clone()

    command = vel_command_b

    @property
    def vel_command_b_buffer(self):
        return self._env.vec.field("LT_F_CMD_BUF")[:, 0, :3].clone()

    @property
    def is_standing_env(self):
        return self._env.vec.field("LT_F_CMD_BUF")[:, 0, 3] != 0

    @property
    def time_left(self):
        return self._env.vec.field("LT_F_CMD")[:, 0, 3].clone()

    @property
    def ranges(self):
        """Current command ranges as the (3, 2) tensor of (lo, hi) rows: lin_vel_x, lin_vel_y, ang_vel_z (LT_F_CMD_PARAMS[0..5])."""
        return self._env.vec.cmd_params[:6].reshape(3, 2).clone()

    @property
    def previous_ranges(self):
        return self._env.vec.cmd_params[6:12].reshape(3, 2).clone()

    @property
    def initial_zero_command_steps(self):
        return int(self._env.vec.cmd_params[15])


class _Commands:
    def __init__(self, env):
        self._env = env
        self._term = _CommandTermView(env)

    def get_command(self, name: str) -> torch.Tensor:
        return self._term.vel_command_b

    def get_term(self, name: str) -> _CommandTermView:
        if name != "base_velocity":
            raise KeyError(f"command term {name!r}: the env has `base_velocity` only")
        return self._term

    active_terms = ["base_velocity"]


class _ActionTermView:
    """`env.action_manager.get_term("joint_pos")`: JointPositionActionPrevPrev's buffers (mdp/actions.py:13-52).  `raw_actions` is
    what the reference's term holds after `process_actions`: clamp(a, +-clip) * raw_action_scale (:39-44) = LT_F_ACT_RAW."""

    def __init__(self, env):
        self._env = env

    def _j(self, name):
        return self._env.vec.field(name).reshape(self._env.num_envs, 12).clone()

    @property
    def raw_actions(self):
        return self._j("LT_F_ACT_RAW")

    @property
    def prev_raw_actions(self):
        return self._j("LT_F_ACT_PREV_RAW")

    @property
    def prev_prev_raw_actions(self):
        return self._j("LT_F_ACT_PREV_PREV_RAW")

    def _processed(self, raw):  # scale 1.0, default offset (locomotion_base_env_cfg.py:126-135)
        return raw + self._env.scene["robot"].data.default_joint_pos

    @property
    def processed_actions(self):
        return self._processed(self.raw_actions)

    @property
    def prev_processed_actions(self):
        return self._processed(self.prev_raw_actions)

    @property
    def prev_prev_processed_actions(self):
        return self._processed(self.prev_prev_raw_actions)

    action_dim = 12


class _Actions:
    def __init__(self, env):
        self._env = env
        self._term = _ActionTermView(env)

    def get_term(self, name: str) -> _ActionTermView:
        if name != "joint_pos":
            raise KeyError(f"action term {name!r}: the env has `joint_pos` only")
        return self._term

    @property
    def action(self):  # the ActionManager's own copy of what the policy sent, up to the clip: raw / raw_action_scale
        return self._term.raw_actions / float(self._env.vec.cfg.action_scale)

    @property
    def prev_action(self):
        return self._term.prev_raw_actions / float(self._env.vec.cfg.action_scale)

    total_action_dim = 12
    active_terms = ["joint_pos"]


class _GaitState:
    """State of the fused gait class as the reference's attributes (rewards.py:96-105), in its foot column order
    [FR, RL, FL, RR]: what `UniformVelocityCommandGaitLogging._update_metrics` reads through
    `reward_manager.get_term_cfg("gait").func.valid_last_air_time` (commands.py:399-403)."""

    def __init__(self, env):
        self._env = env

    def _feet(self, name):
        return self._env.vec.field(name)[:, 0, :][:, GAIT_FOOT_ORDER].clone()

    @property
    def valid_last_air_time(self):
        return self._feet("LT_F_GAIT_VALID_LAST_AIR")

    @property
    def last_step_current_air_time(self):
        return self._feet("LT_F_GAIT_LAST_AIR")

    @property
    def last_step_current_contact_time(self):
        return self._feet("LT_F_GAIT_LAST_CONTACT")

    @property
    def last_velocity_cmd(self):
        return self._env.vec.field("LT_F_GAIT_CMD")[:, 0, :3].clone()

    @property
    def step_from_changing_cmd(self):
        return self._env.vec.field("LT_F_GAIT_CMD")[:, 0, 3].clone()

    def __call__(self, *a, **k):
        raise RuntimeError("the gait reward is evaluated inside lt_step_kernel; this object only exposes its state")


class _TermCfg:
    def __init__(self, func, weight, params):
        self.func, self.weight, self.params = func, float(weight), dict(params)


def _fused_reward_params(cfg) -> dict:
    """`params` of the fused reward terms that other terms reach into (rewards.py:380-381: the gait-with-object class reads
    `object_dangerous_state`'s x_max / y_max; curriculums.py:197-200: the curriculum reads the tracking terms' sigma / weight)."""
    return {
        "track_lin_vel_xy": {"sigma": float(cfg.track_sigma)}, "track_ang_vel_z": {"sigma": float(cfg.track_sigma)},
        "foot_slip": {"threshold": float(cfg.foot_slip_threshold)},
        "foot_dragging": {"height_threshold": float(cfg.foot_drag_height), "foot_vel_xy_threshold": float(cfg.foot_drag_vel)},
        "track_base_height": {"target_height": float(cfg.base_height_target)},
        "joint_position": {"stand_still_scale": float(cfg.joint_pos_stand_scale), "velocity_threshold": float(cfg.joint_pos_vel_threshold)},
        "thigh_calf_collision": {"threshold": float(cfg.thigh_calf_threshold)},
        "object_dangerous_state": {"x_max": float(cfg.danger_x_max), "y_max": float(cfg.danger_y_max), "z_min": float(cfg.danger_z_min),
                                   "roll_pitch_max": None, "vel_xy_max": float(cfg.danger_vel_xy_max)},
    }


class _EpisodeSums:
    """`reward_manager._episode_sums[name]` -> (N,) running weighted sum of a term over the current episode
    (RewardManager [DEP]; read by curriculums.py:238,259): the kernel's LT_F_EPISODE_SUMS column, or the slow path's own sum."""

    def __init__(self, env):
        self._env = env

    def __getitem__(self, name):
        extra = self._env.extra_sums
        if extra is not None and name in extra:
            return extra[name]
        i = REWARD_TERM_NAMES.index(name) if name in REWARD_TERM_NAMES else None
        if i is None:
            raise KeyError(name)
        return self._env.vec.field("LT_F_EPISODE_SUMS").reshape(self._env.num_envs, -1)[:, i].clone()

    def __contains__(self, name):
        return name in REWARD_TERM_NAMES or (self._env.extra_sums is not None and name in self._env.extra_sums)

    def keys(self):
        return [n for n in REWARD_TERM_NAMES if self._env.vec.cfg.reward_weight[REWARD_TERM_NAMES.index(n)] != 0] + \
            (list(self._env.extra_sums) if self._env.extra_sums is not None else [])


class _Rewards:
    def __init__(self, env):
        self._env = env
        self._episode_sums = _EpisodeSums(env)
        self._gait = _GaitState(env)
        self._fused_params = _fused_reward_params(env.vec.cfg)
        self.user_cfgs: dict = {}  # name -> _TermCfg of the slow-path terms (ExtraTerms.add_reward)

    def get_term_cfg(self, name: str) -> _TermCfg:
        if name in self.user_cfgs:
            return self.user_cfgs[name]
        if name not in REWARD_TERM_NAMES:
            raise ValueError(f"reward term {name!r} not found")
        w = float(self._env.vec.cfg.reward_weight[REWARD_TERM_NAMES.index(name)])
        func = self._gait if name == "gait" else _fused_placeholder(name)
        return _TermCfg(func, w, self._fused_params.get(name, {}))

    @property
    def active_terms(self):
        return self._episode_sums.keys()


def _fused_placeholder(name):
    def term(env, *a, **k):
        raise RuntimeError(f"reward term {name!r} is evaluated inside lt_step_kernel (its unweighted value: LT_F_REWARD_TERMS with cfg.debug_terms)")

    term.__name__ = name
    return term


class _Terminations:
    """`env.termination_manager`: terminated / time_outs / dones of the last step (TerminationManager [DEP]) and per-term bits."""

    def __init__(self, env):
        self._env = env

    @property
    def terminated(self):
        return self._env.vec.field("LT_F_TERMINATED") != 0

    @property
    def time_outs(self):
        return self._env.vec.field("LT_F_TIME_OUT") != 0

    @property
    def dones(self):
        return self._env.vec.field("LT_F_DONES") != 0

    def get_term(self, name: str) -> torch.Tensor:
        names = TERMINATION_NAMES + ["user", "user_time_out"]
        return ((self._env.vec.field("LT_F_TERM_BITS") >> names.index(name)) & 1).bool()

    @property
    def active_terms(self):
        return [n for b, n in enumerate(TERMINATION_NAMES) if self._env.vec.cfg.term_enabled[b]]


def link_kinematics(p0, q0, v0, w0, jq, jqd):
    """World pose and velocity of the 17 sensor bodies [trunk, 4 hips, 4 thighs, 4 calves, 4 feet] (IsaacLab's breadth-first body
    order) from the root state and the joint state (N, 12) in joint order type * 4 + leg.
    Returns pos (N,17,3), quat wxyz (N,17,4), lin vel (N,17,3), ang vel (N,17,3).  Plain forward kinematics of the URDF tree:
    hip about x, thigh and calf about y, foot fixed to the calf (include/lt_go1_model.h)."""
    n, dev = p0.shape[0], p0.device
    t = lambda x: torch.tensor(x, dtype=p0.dtype, device=dev)  # noqa: E731
    pos, quat, lin, ang = [p0], [q0], [v0], [w0]
    per_type: list[list] = [[], [], [], []]
    ex, ey = t([1.0, 0.0, 0.0]).expand(n, 3), t([0.0, 1.0, 0.0]).expand(n, 3)
    zero = torch.zeros(n, dtype=p0.dtype, device=dev)

    def axis_quat(angle, axis):  # rotation about x (0) or y (1)
        h = 0.5 * angle
        c, s = torch.cos(h), torch.sin(h)
        return torch.stack((c, s, zero, zero) if axis == 0 else (c, zero, s, zero), dim=-1)

    for leg in range(4):
        pp, qq, vv, ww = p0, q0, v0, w0
        for k, (off, axis) in enumerate(((HIP_OFFSET[leg], 0), (THIGH_OFFSET[leg], 1), (CALF_OFFSET, 1))):
            r = M.quat_apply(qq, t(off).expand(n, 3))          # joint origin relative to the parent origin, world axes
            vv = vv + torch.cross(ww, r, dim=-1)
            pp = pp + r
            a = M.quat_apply(qq, ex if axis == 0 else ey)      # joint axis in the world (the same in parent and child frame)
            ww = ww + a * jqd[:, k * 4 + leg: k * 4 + leg + 1]
            qq = M.quat_mul(qq, axis_quat(jq[:, k * 4 + leg], axis))
            per_type[k].append((pp, qq, vv, ww))
        r = M.quat_apply(qq, t(FOOT_OFFSET).expand(n, 3))
        per_type[3].append((pp + r, qq, vv + torch.cross(ww, r, dim=-1), ww))
    for k in range(4):
        for leg in range(4):
            a, b, c, d = per_type[k][leg]
            pos.append(a), quat.append(b), lin.append(c), ang.append(d)
    return torch.stack(pos, 1), torch.stack(quat, 1), torch.stack(lin, 1), torch.stack(ang, 1)


class TermEnv:
    """What a manager term function receives as `env`: scene / command_manager / action_manager / reward_manager /
    termination_manager / step_dt / num_envs / device / episode_length_buf / max_episode_length(_s) / common_step_counter over a
    VecEnv with `field(name)` quad views (LocoTouchVecEnv, or the test oracle env)."""

    def __init__(self, vec):
        self.vec = vec
        self.num_envs, self.device = vec.num_envs, vec.device
        self.physics_dt = float(vec.cfg.sim_dt)
        self.step_dt = float(vec.cfg.sim_dt) * int(vec.cfg.decimation)
        self.max_episode_length = int(vec.cfg.max_episode_length)
        self.max_episode_length_s = float(vec.cfg.episode_length_s)
        self.extra_sums: dict | None = None  # ExtraTerms.sums (slow-path reward terms)
        self.cmd_frozen = None  # ExtraTerms.pre_step: the command as the step's termination / reward stages see it
        n = self.num_envs
        f = vec.field
        has_obj = int(vec.cfg.task) != _abi.CONSTS["LT_TASK_LOCOMOTION"]
        v3 = lambda name: (lambda: f(name)[:, 0, :3].clone())  # noqa: E731
        quat = lambda name: (lambda: f(name)[:, 0, :4].clone())  # noqa: E731
        j12 = lambda name: (lambda: f(name).reshape(n, 12).clone())  # noqa: E731  (component = type * 4 + leg: IsaacLab's breadth-first joint order)

        def body(pose_q, vec_w):
            return lambda: M.quat_apply_inverse(f(pose_q)[:, 0, :4], f(vec_w)[:, 0, :3])

        def gravity(pose_q):
            return lambda: M.quat_apply_inverse(f(pose_q)[:, 0, :4], torch.tensor([0.0, 0.0, -1.0], device=self.device).expand(n, 3))

        dq = torch.tensor([-0.1, 0.1, -0.1, 0.1, 0.9, 0.9, 0.9, 0.9, -1.8, -1.8, -1.8, -1.8], device=self.device)  # assets/go1.py:31-38
        lo = torch.tensor([-0.863] * 4 + [-0.686] * 4 + [-2.818] * 4, device=self.device)
        hi = torch.tensor([0.863] * 4 + [4.501] * 4 + [-0.888] * 4, device=self.device)
        mid, rng = (lo + hi) / 2, hi - lo
        soft = torch.stack((mid - 0.5 * rng * 0.95, mid + 0.5 * rng * 0.95), dim=-1)  # soft_joint_pos_limit_factor 0.95 (go1.py:29)
        hard = torch.stack((lo, hi), dim=-1)

        def root_state():
            return torch.cat([f("LT_F_ROOT_POS")[:, 0, :3], f("LT_F_ROOT_QUAT")[:, 0, :4], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3],
                              f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3]], dim=1)

        def links():
            out = link_kinematics(f("LT_F_ROOT_POS")[:, 0, :3], f("LT_F_ROOT_QUAT")[:, 0, :4], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3],
                                  f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3], f("LT_F_JOINT_POS").reshape(n, 12), f("LT_F_JOINT_VEL").reshape(n, 12))
            # the feet are the kernel's own foot kinematics (what the fused foot_slip / foot_dragging terms read)
            out[0][:, 13:17] = f("LT_F_FOOT_POS_W").permute(0, 2, 1)
            out[2][:, 13:17] = f("LT_F_FOOT_VEL_W").permute(0, 2, 1)
            return out

        default_root = torch.tensor([0.0, 0.0, 0.28, 1.0, 0.0, 0.0, 0.0] + [0.0] * 6, device=self.device)  # assets/go1.py:30-31
        robot = _Data({
            "root_pos_w": v3("LT_F_ROOT_POS"), "root_quat_w": quat("LT_F_ROOT_QUAT"), "root_lin_vel_w": v3("LT_F_ROOT_LIN_VEL_W"),
            "root_ang_vel_w": v3("LT_F_ROOT_ANG_VEL_W"), "root_lin_vel_b": body("LT_F_ROOT_QUAT", "LT_F_ROOT_LIN_VEL_W"),
            "root_ang_vel_b": body("LT_F_ROOT_QUAT", "LT_F_ROOT_ANG_VEL_W"), "projected_gravity_b": gravity("LT_F_ROOT_QUAT"),
            "joint_pos": j12("LT_F_JOINT_POS"), "joint_vel": j12("LT_F_JOINT_VEL"), "joint_acc": j12("LT_F_JOINT_ACC"),
            "applied_torque": j12("LT_F_APPLIED_TORQUE"), "default_joint_pos": lambda: dq.expand(n, 12).clone(),
            "default_joint_vel": lambda: torch.zeros(n, 12, device=self.device), "soft_joint_pos_limits": lambda: soft.expand(n, 12, 2).clone(),
            "joint_pos_limits": lambda: hard.expand(n, 12, 2).clone(), "joint_limits": lambda: hard.expand(n, 12, 2).clone(),
            "root_state_w": root_state, "default_root_state": lambda: default_root.expand(n, 13).clone(),
            "body_pos_w": lambda: links()[0], "body_quat_w": lambda: links()[1], "body_lin_vel_w": lambda: links()[2],
            "body_ang_vel_w": lambda: links()[3],
            "body_state_w": lambda: torch.cat(links(), dim=-1),
        })

        def force_norms():  # (N, 3 slots, 17 bodies): trunk, then body 1 + type * 4 + leg
            fh = f("LT_F_FORCE_HIST").reshape(n, 3, 4, 4)  # [slot][type][leg]
            tr = f("LT_F_TRUNK_FORCE_HIST")[:, 0, :3]
            return torch.cat([tr.unsqueeze(-1), fh.reshape(n, 3, 16)], dim=-1)

        def forces_hist():
            out = torch.zeros(n, 3, 17, 3, device=self.device)
            out[..., 2] = force_norms()
            return out

        def feet(name):
            def g():
                out = torch.zeros(n, 17, device=self.device)
                out[:, 13:17] = f(name)[:, 0, :4]
                return out
            return g

        contact = _Data({
            "net_forces_w_history": forces_hist, "net_forces_w": lambda: forces_hist()[:, 0], "force_norm_history": force_norms,
            "current_air_time": feet("LT_F_FOOT_CUR_AIR"), "current_contact_time": feet("LT_F_FOOT_CUR_CONTACT"),
            "last_air_time": feet("LT_F_FOOT_LAST_AIR"), "last_contact_time": feet("LT_F_FOOT_LAST_CONTACT"),
        })
        ents = {"robot": _Entity(robot, BODY_NAMES, JOINT_NAMES)}
        robot_sensor = _Entity(contact, BODY_NAMES)
        sensors = {ROBOT_SENSOR: robot_sensor, "contact_forces": robot_sensor}  # (the second: the stock IsaacLab velocity-task name)
        if has_obj:
            ents["object"] = _Entity(_Data({
                "root_pos_w": v3("LT_F_OBJ_POS"), "root_quat_w": quat("LT_F_OBJ_QUAT"), "root_lin_vel_w": v3("LT_F_OBJ_LIN_VEL_W"),
                "root_ang_vel_w": v3("LT_F_OBJ_ANG_VEL_W"), "projected_gravity_b": gravity("LT_F_OBJ_QUAT"),
                "root_lin_vel_b": body("LT_F_OBJ_QUAT", "LT_F_OBJ_LIN_VEL_W"), "root_ang_vel_b": body("LT_F_OBJ_QUAT", "LT_F_OBJ_ANG_VEL_W"),
                "root_state_w": lambda: torch.cat([f("LT_F_OBJ_POS")[:, 0, :3], f("LT_F_OBJ_QUAT")[:, 0, :4], f("LT_F_OBJ_LIN_VEL_W")[:, 0, :3],
                                                   f("LT_F_OBJ_ANG_VEL_W")[:, 0, :3]], dim=1),
            }), ["Object"])
            ot = lambda i: (lambda: f("LT_F_OBJ_TIMERS")[:, 0, i:i + 1].clone())  # noqa: E731
            sensors["object_contact_sensor"] = _Entity(_Data({"current_air_time": ot(0), "current_contact_time": ot(1), "last_air_time": ot(2),
                                                              "last_contact_time": ot(3)}), ["Object"])
        self.scene = Scene(ents, sensors, n)
        self.command_manager = _Commands(self)
        self.action_manager = _Actions(self)
        self.reward_manager = _Rewards(self)
        self.termination_manager = _Terminations(self)

    @property
    def episode_length_buf(self):
        return self.vec.episode_length_buf

    @property
    def common_step_counter(self) -> int:
        """Env steps taken so far (ManagerBasedRLEnv.common_step_counter [DEP]): LT_F_COUNTERS[0]; reading it waits for the stream."""
        return int(self.vec.field("LT_F_COUNTERS")[0])

    @property
    def unwrapped(self):
        return self


class ExtraTerms:
    """Reward and termination terms outside the fused set, evaluated in torch on `TermEnv` after every step (module docstring)."""

    def __init__(self, vec):
        self.env = TermEnv(vec)
        self.vec = vec
        self.terms: list = []  # (name, callable, weight, params)
        self.sums: dict = {}
        self.env.extra_sums = self.sums
        self.terminations: list = []  # (name, callable, params, time_out)
        self.term_counts: dict = {}   # name -> envs terminated by the term since the last episode_log read
        self.last_values: dict = {}   # name -> the unweighted value of the last step (diagnostics / parity tests)
        self.observations: list = []  # user observation terms (add_observation)

    def pre_step(self) -> None:
        """Call before the env step.  The reference computes terminations and rewards BEFORE the command term's update of the same
        step (stages 4-5 vs 7), i.e. with the command the step started with; the arena holds the updated command
        after the step, so the terms are served this copy."""
        self.env.cmd_frozen = self.vec.field("LT_F_CMD")[:, 0, :3].clone()

    def post_step(self) -> None:
        self.env.cmd_frozen = None

    def _bind(self, func, params, weight=0.0):
        params = dict(params or {})
        for v in params.values():  # SceneEntityCfg-like parameters: names -> ids against this scene (the manager does that at load [DEP])
            if hasattr(v, "resolve") and hasattr(v, "name"):
                v.resolve(self.env.scene)
        cfg = _TermCfg(func, weight, params)
        if isinstance(func, type):  # class term (ManagerTermBase): built with (cfg, env), called like a function
            func = func(cfg, self.env)
            cfg.func = func
        return func, params, cfg

    def add_termination(self, name: str, func, params: dict | None = None, time_out: bool = False) -> None:
        """A termination term `func(env, **params) -> bool (N,)` (reference signature, mdp/terminations.py:10-23); `time_out`:
        the TerminationTermCfg flag - the env ends by time-out (bootstrapped by PPO, ppo.py:162-165) instead of terminating."""
        func, params, _ = self._bind(func, params)
        self.terminations.append((name, func, params, bool(time_out)))
        self.term_counts[name] = 0

    def request_terminations(self, dones: torch.Tensor) -> torch.Tensor:
        """Evaluate the user termination terms on the state the step left; envs they fire for (and that did not just finish)
        are ended by the NEXT step (LocoTouchVecEnv.request_termination).  Returns the mask."""
        n, dev = self.env.num_envs, self.env.device
        fired = torch.zeros(n, dtype=torch.bool, device=dev)
        timed = torch.zeros(n, dtype=torch.bool, device=dev)
        alive = dones == 0
        for name, func, params, time_out in self.terminations:
            m = func(self.env, **params).to(torch.bool).reshape(n) & alive
            self.term_counts[name] += int(m.sum())
            if time_out:
                timed |= m
            else:
                fired |= m
        if bool(fired.any()):
            self.vec.request_termination(fired)
        if bool(timed.any()):
            self.vec.request_termination(timed, time_out=True)
        return fired | timed

    def add_reward(self, name: str, func, weight: float, params: dict | None = None) -> None:
        func, params, cfg = self._bind(func, params, weight)
        self.terms.append((name, func, float(weight), params))
        self.sums[name] = torch.zeros(self.env.num_envs, device=self.env.device)
        self.env.reward_manager.user_cfgs[name] = cfg

    def __bool__(self) -> bool:
        return bool(self.terms) or bool(self.terminations) or bool(self.observations)

    # ---- user observation terms (ObservationManager.compute_group [DEP]: func -> noise (if the group corrupts) -> clip -> scale ->
    #      history buffer, flattened oldest -> newest; a reset env's buffer is filled with its first value) -------------------------
    def add_observation(self, group: str, name: str, func, params: dict | None = None, history_length: int = 0, scale=None, clip=None,
                        noise=None) -> None:
        """An observation term `func(env, **params) -> (N, d)` appended to `group` ("policy" / "critic") behind the fused terms.
        `noise`: (n_min, n_max) of an additive uniform model, applied to this group (the caller passes it for a corrupting group only)."""
        func, params, _ = self._bind(func, params)
        self.observations.append(dict(group=group, name=name, func=func, params=params, hist=int(history_length or 0), scale=scale, clip=clip,
                                      noise=noise, buf=None, fresh=None))

    def _term_value(self, o) -> torch.Tensor:
        v = o["func"](self.env, **o["params"]).to(torch.float32).reshape(self.env.num_envs, -1).clone()
        if o["noise"] is not None:
            lo, hi = o["noise"]
            v = v + torch.rand_like(v) * (hi - lo) + lo
        if o["clip"] is not None:
            v = v.clamp(float(o["clip"][0]), float(o["clip"][1]))
        if o["scale"] is not None:
            v = v * (torch.as_tensor(o["scale"], dtype=v.dtype, device=v.device) if not isinstance(o["scale"], (int, float)) else float(o["scale"]))
        return v

    def observe(self, dones: torch.Tensor | None) -> dict:
        """{group: (N, sum of d * history)} of the user observation terms for the state the env is in now; `dones` (None: every env
        starts an episode - after reset()): the envs whose history restarts with this value."""
        out: dict = {}
        n = self.env.num_envs
        for o in self.observations:
            v = self._term_value(o)
            h = max(1, o["hist"])
            if o["buf"] is None or dones is None:
                o["buf"] = v.unsqueeze(1).repeat(1, h, 1)
            else:
                o["buf"] = torch.cat((o["buf"][:, 1:], v.unsqueeze(1)), dim=1)
                fin = (dones != 0).reshape(n)
                if bool(fin.any()):
                    o["buf"][fin] = v[fin].unsqueeze(1)
            out.setdefault(o["group"], []).append(o["buf"].reshape(n, -1))
        return {g: torch.cat(vs, dim=1) for g, vs in out.items()}

    def observation_dims(self) -> dict:
        dims: dict = {}
        for o in self.observations:
            d = int(self._term_value(o).shape[1]) * max(1, o["hist"])
            dims[o["group"]] = dims.get(o["group"], 0) + d
        return dims

    def apply(self, reward: torch.Tensor, dones: torch.Tensor) -> torch.Tensor:
        """reward + sum_i weight_i * dt * term_i(env) for the envs that did not just finish (in place on a copy of `reward`)."""
        keep = (dones == 0).to(reward.dtype)
        out = reward.clone()
        for name, func, w, params in self.terms:
            if w == 0.0:
                continue  # RewardManager: a zero-weight term is not evaluated [DEP]
            raw = func(self.env, **params).to(reward.dtype).reshape(self.env.num_envs)
            self.last_values[name] = raw
            v = raw * (w * self.env.step_dt) * keep
            out += v
            self.sums[name] = (self.sums[name] + v) * keep
        finished = dones != 0
        if bool(finished.any()):  # class terms: ManagerTermBase.reset(env_ids) on the envs that finished (RewardManager.reset [DEP])
            ids = finished.nonzero(as_tuple=True)[0]
            for _, func, _, _ in self.terms:
                if hasattr(func, "reset") and not isinstance(func, type):
                    func.reset(ids)
        return out
