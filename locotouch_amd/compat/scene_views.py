"""IsaacLab-layout views of the env state for manager terms the fused kernels do not know.

The reference's terms read `env.scene["robot"].data.joint_pos`, `env.scene.sensors["contact_forces"].data.net_forces_w_history`,
`env.command_manager.get_command("base_velocity")`, ... (locotouch/mdp/rewards.py:39,423-456,464; events.py:174-196).  The HIP env
keeps its state in quad arrays (include/lt_layout.h); `TermEnv` presents them in the reference's layouts - torch tensors built from
the arena views on every access, on the env's device - and `ExtraTerms` evaluates user reward terms on them after a step and adds
`weight * dt * value` to the reward the kernel computed (the SLOW path: a handful of torch launches per term and step).

Limits, stated: the step kernel resets finished envs inside the step, so a user term sees the post-reset state of an env that just
finished and its value is dropped there (the fused terms are evaluated before the reset, as the RewardManager does [DEP]);
contact forces are stored as norms - `net_forces_w_history` carries each body's |F| in the z component (norm-exact, direction not
kept); air / contact timers exist for the four feet.  User TERMINATION terms (time_out = False) are evaluated on the state a step
left and take effect through the kernel's own termination stage one env step LATER (LT_T_USER in include/lt_env.h: the reset has to
happen inside the step kernel); user time-out terms still raise UnsupportedCfg.
"""
from __future__ import annotations

import re

import torch

from .. import _abi
from . import math as M

BODY_NAMES = ["trunk"] + [f"{leg}_{part}" for part in ("hip", "thigh", "calf", "foot") for leg in ("a_FR", "b_FL", "c_RR", "d_RL")]
JOINT_NAMES = [f"{leg}_{part}_joint" for part in ("hip", "thigh", "calf") for leg in ("a_FR", "b_FL", "c_RR", "d_RL")]


def _find(keys, names, preserve_order=False):
    keys = [keys] if isinstance(keys, str) else list(keys)
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


class _Commands:
    def __init__(self, env):
        self._env = env

    def get_command(self, name: str) -> torch.Tensor:
        return self._env.vec.field("LT_F_CMD")[:, 0, :3].clone()


class _Actions:
    def __init__(self, env):
        self._env = env

    @property
    def action(self):
        s = float(self._env.vec.cfg.action_scale)
        return self._env.vec.field("LT_F_ACT_RAW").reshape(self._env.num_envs, 12) / s

    @property
    def prev_action(self):
        s = float(self._env.vec.cfg.action_scale)
        return self._env.vec.field("LT_F_ACT_PREV_RAW").reshape(self._env.num_envs, 12) / s


class TermEnv:
    """What a manager term function receives as `env`: scene / command_manager / action_manager / step_dt / num_envs / device /
    episode_length_buf / max_episode_length over a VecEnv with `field(name)` quad views (LocoTouchVecEnv, or the test oracle env)."""

    def __init__(self, vec):
        self.vec = vec
        self.num_envs, self.device = vec.num_envs, vec.device
        self.step_dt = float(vec.cfg.sim_dt) * int(vec.cfg.decimation)
        self.max_episode_length = int(vec.cfg.max_episode_length)
        self.max_episode_length_s = self.max_episode_length * self.step_dt
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

        mirror = _abi.CONSTS
        dq = torch.tensor([-0.1, 0.1, -0.1, 0.1, 0.9, 0.9, 0.9, 0.9, -1.8, -1.8, -1.8, -1.8], device=self.device)  # assets/go1.py:31-38
        lo = torch.tensor([-0.863] * 4 + [-0.686] * 4 + [-2.818] * 4, device=self.device)
        hi = torch.tensor([0.863] * 4 + [4.501] * 4 + [-0.888] * 4, device=self.device)
        mid, rng = (lo + hi) / 2, hi - lo
        soft = torch.stack((mid - 0.5 * rng * 0.95, mid + 0.5 * rng * 0.95), dim=-1)  # soft_joint_pos_limit_factor 0.95 (go1.py:29)
        _ = mirror
        robot = _Data({
            "root_pos_w": v3("LT_F_ROOT_POS"), "root_quat_w": quat("LT_F_ROOT_QUAT"), "root_lin_vel_w": v3("LT_F_ROOT_LIN_VEL_W"),
            "root_ang_vel_w": v3("LT_F_ROOT_ANG_VEL_W"), "root_lin_vel_b": body("LT_F_ROOT_QUAT", "LT_F_ROOT_LIN_VEL_W"),
            "root_ang_vel_b": body("LT_F_ROOT_QUAT", "LT_F_ROOT_ANG_VEL_W"), "projected_gravity_b": gravity("LT_F_ROOT_QUAT"),
            "joint_pos": j12("LT_F_JOINT_POS"), "joint_vel": j12("LT_F_JOINT_VEL"), "joint_acc": j12("LT_F_JOINT_ACC"),
            "applied_torque": j12("LT_F_APPLIED_TORQUE"), "default_joint_pos": lambda: dq.expand(n, 12).clone(),
            "default_joint_vel": lambda: torch.zeros(n, 12, device=self.device), "soft_joint_pos_limits": lambda: soft.expand(n, 12, 2).clone(),
            "root_state_w": lambda: torch.cat([f("LT_F_ROOT_POS")[:, 0, :3], f("LT_F_ROOT_QUAT")[:, 0, :4], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3],
                                               f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3]], dim=1),
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
        sensors = {"contact_forces": _Entity(contact, BODY_NAMES)}
        if has_obj:
            ents["object"] = _Entity(_Data({
                "root_pos_w": v3("LT_F_OBJ_POS"), "root_quat_w": quat("LT_F_OBJ_QUAT"), "root_lin_vel_w": v3("LT_F_OBJ_LIN_VEL_W"),
                "root_ang_vel_w": v3("LT_F_OBJ_ANG_VEL_W"), "projected_gravity_b": gravity("LT_F_OBJ_QUAT"),
            }), ["Object"])
            ot = lambda i: (lambda: f("LT_F_OBJ_TIMERS")[:, 0, i:i + 1].clone())  # noqa: E731
            sensors["object_contact_sensor"] = _Entity(_Data({"current_air_time": ot(0), "current_contact_time": ot(1), "last_air_time": ot(2),
                                                              "last_contact_time": ot(3)}), ["Object"])
        self.scene = Scene(ents, sensors, n)
        self.command_manager = _Commands(self)
        self.action_manager = _Actions(self)

    @property
    def episode_length_buf(self):
        return self.vec.episode_length_buf


class ExtraTerms:
    """Reward terms outside the fused set, evaluated in torch on `TermEnv` after every step (module docstring)."""

    def __init__(self, vec):
        self.env = TermEnv(vec)
        self.vec = vec
        self.terms: list = []  # (name, callable, weight, params)
        self.sums: dict = {}
        self.terminations: list = []  # (name, callable, params)
        self.term_counts: dict = {}   # name -> envs terminated by the term since the last episode_log read

    def _bind(self, func, params, weight=0.0):
        params = dict(params or {})
        for v in params.values():  # SceneEntityCfg-like parameters: names -> ids against this scene (the manager does that at load [DEP])
            if hasattr(v, "resolve") and hasattr(v, "name"):
                v.resolve(self.env.scene)
        if isinstance(func, type):  # class term (ManagerTermBase): built with (cfg, env), called like a function
            cfg = type("Cfg", (), {"params": params, "weight": weight, "func": func})()
            func = func(cfg, self.env)
        return func, params

    def add_termination(self, name: str, func, params: dict | None = None) -> None:
        """A termination term `func(env, **params) -> bool (N,)` (reference signature, mdp/terminations.py:10-23)."""
        func, params = self._bind(func, params)
        self.terminations.append((name, func, params))
        self.term_counts[name] = 0

    def request_terminations(self, dones: torch.Tensor) -> torch.Tensor:
        """Evaluate the user termination terms on the state the step left; envs they fire for (and that did not just finish)
        are terminated by the NEXT step (LocoTouchVecEnv.request_termination).  Returns the mask."""
        fired = torch.zeros(self.env.num_envs, dtype=torch.bool, device=self.env.device)
        alive = dones == 0
        for name, func, params in self.terminations:
            m = func(self.env, **params).to(torch.bool) & alive
            self.term_counts[name] += int(m.sum())
            fired |= m
        if bool(fired.any()):
            self.vec.request_termination(fired)
        return fired

    def add_reward(self, name: str, func, weight: float, params: dict | None = None) -> None:
        func, params = self._bind(func, params, weight)
        self.terms.append((name, func, float(weight), params))
        self.sums[name] = torch.zeros(self.env.num_envs, device=self.env.device)

    def __bool__(self) -> bool:
        return bool(self.terms) or bool(self.terminations)

    def apply(self, reward: torch.Tensor, dones: torch.Tensor) -> torch.Tensor:
        """reward + sum_i weight_i * dt * term_i(env) for the envs that did not just finish (in place on a copy of `reward`)."""
        keep = (dones == 0).to(reward.dtype)
        out = reward.clone()
        for name, func, w, params in self.terms:
            if w == 0.0:
                continue  # RewardManager: a zero-weight term is not evaluated [DEP]
            v = func(self.env, **params).to(reward.dtype) * (w * self.env.step_dt) * keep
            out += v
            self.sums[name] = (self.sums[name] + v) * keep
        return out
