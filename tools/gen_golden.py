#!/usr/bin/env python3
"""Golden-vector generator for the `locotouch/mdp` terms.

Runs ONLY in the build container: it imports the reference's own `locotouch.mdp` source from
/root/reference (read-only, never copied) on top of the throw-away `isaaclab` stand-in
(locotouch_amd/compat/isaaclab_shim.py) and drives each term with a duck-typed fake env holding
seeded synthetic state.  What is written to tests/golden/*.npz is data only: the synthetic inputs
(in IsaacLab's data-contract layout) and the tensors the reference functions
returned for them.

Caveat recorded in DESIGN.md: the seven `isaaclab.utils.math` helpers and the manager base classes
used underneath are this repo's restatement of IsaacLab (absent here) - parity is pinned at the
`locotouch.mdp` level, unpinned at the IsaacLab boundary.

    python tools/gen_golden.py            # writes tests/golden/mdp_*.npz
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from locotouch_amd.compat import isaaclab_shim
from locotouch_amd.compat import math as M

isaaclab_shim.install(import_subpackages=False)
sys.path.insert(0, "/root/reference")
import locotouch.mdp as mdp  # noqa: E402  (the reference's own source)
from isaaclab.managers import SceneEntityCfg, RewardTermCfg, CurriculumTermCfg  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
BODY_NAMES = (["trunk"] + [f"{l}_hip" for l in ("a_FR", "b_FL", "c_RR", "d_RL")]
              + [f"{l}_thigh" for l in ("a_FR", "b_FL", "c_RR", "d_RL")]
              + [f"{l}_calf" for l in ("a_FR", "b_FL", "c_RR", "d_RL")]
              + [f"{l}_foot" for l in ("a_FR", "b_FL", "c_RR", "d_RL")])
JOINT_NAMES = [f"{l}_{k}_joint" for k in ("hip", "thigh", "calf") for l in ("a_FR", "b_FL", "c_RR", "d_RL")]
DEFAULT_Q = torch.tensor([-0.1, 0.1, -0.1, 0.1] + [0.9] * 4 + [-1.8] * 4)
LOWER = torch.tensor([-0.863] * 4 + [-0.686] * 4 + [-2.818] * 4)
UPPER = torch.tensor([0.863] * 4 + [4.501] * 4 + [-0.888] * 4)
STEP_DT = 0.02


class Data(types.SimpleNamespace):
    pass


class FakeAsset:
    def __init__(self, names=None, joint_names=None):
        self.data = Data()
        self.body_names = names or []
        self.joint_names = joint_names or []
        self.device = "cpu"
        self.cfg = types.SimpleNamespace(spawn=None)

    @property
    def num_bodies(self):
        return len(self.body_names)

    @property
    def num_joints(self):
        return len(self.joint_names)

    def find_bodies(self, keys, preserve_order=False):
        return isaaclab_shim.resolve_matching_names(keys, self.body_names, preserve_order)

    def find_joints(self, keys, preserve_order=False):
        return isaaclab_shim.resolve_matching_names(keys, self.joint_names, preserve_order)


class FakeScene(dict):
    def __init__(self, n):
        super().__init__()
        self.num_envs = n
        self.sensors = {}

    def __getitem__(self, k):
        if k in self.sensors:
            return self.sensors[k]
        return dict.__getitem__(self, k)


class FakeEnv:
    def __init__(self, n, max_episode_length_s=20.0):
        self.num_envs = n
        self.device = "cpu"
        self.step_dt = STEP_DT
        self.max_episode_length_s = max_episode_length_s
        self.max_episode_length = math.ceil(max_episode_length_s / STEP_DT)
        self.episode_length_buf = torch.zeros(n, dtype=torch.long)
        self.scene = FakeScene(n)
        self.scene["robot"] = FakeAsset(BODY_NAMES, JOINT_NAMES)
        self.scene["object"] = FakeAsset(["Object"])
        self.scene.sensors["robot_contact_senosr"] = FakeAsset(BODY_NAMES)
        self.scene.sensors["object_contact_sensor"] = FakeAsset(["Object"])
        self.cmd = torch.zeros(n, 3)
        self.command_manager = types.SimpleNamespace(get_command=lambda name: self.cmd, get_term=lambda name: self.cmd_term)
        self.cmd_term = None
        self.action_term = types.SimpleNamespace(raw_actions=torch.zeros(n, 12), prev_raw_actions=torch.zeros(n, 12))
        self.action_manager = types.SimpleNamespace(get_term=lambda name: self.action_term)
        self.reward_cfgs = {}
        self.reward_manager = types.SimpleNamespace(get_term_cfg=lambda name: self.reward_cfgs[name], _episode_sums={})
        self.termination_manager = types.SimpleNamespace(terminated=torch.zeros(n, dtype=torch.bool))
        rd = self.scene["robot"].data
        rd.default_joint_pos = DEFAULT_Q.repeat(n, 1)
        mid = (LOWER + UPPER) / 2
        rng = (UPPER - LOWER)
        rd.soft_joint_pos_limits = torch.stack([mid - 0.5 * rng * 0.95, mid + 0.5 * rng * 0.95], dim=-1).repeat(n, 1, 1)


def rand_quat(g, n, rp=0.4):
    roll = (torch.rand(n, generator=g) * 2 - 1) * rp
    pitch = (torch.rand(n, generator=g) * 2 - 1) * rp
    yaw = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    return M.quat_from_euler_xyz(roll, pitch, yaw)


def U(g, shape, lo, hi):
    """Uniform sample snapped to a 2^-10 grid (exact in fp32; keeps the committed .npz small)."""
    return torch.round((torch.rand(*shape, generator=g) * (hi - lo) + lo) * 1024.0) / 1024.0


class ContactTimers:
    """ContactSensor air/contact bookkeeping driven by synthetic contact booleans."""

    def __init__(self, n, b):
        self.cur_air = torch.zeros(n, b)
        self.cur_con = torch.zeros(n, b)
        self.last_air = torch.zeros(n, b)
        self.last_con = torch.zeros(n, b)

    def update(self, contact, dt):
        first_contact = (self.cur_air > 0) & contact
        first_detach = (self.cur_con > 0) & ~contact
        self.last_air = torch.where(first_contact, self.cur_air + dt, self.last_air)
        self.last_con = torch.where(first_detach, self.cur_con + dt, self.last_con)
        self.cur_air = torch.where(~contact, self.cur_air + dt, torch.zeros_like(self.cur_air))
        self.cur_con = torch.where(contact, self.cur_con + dt, torch.zeros_like(self.cur_con))

    def reset(self, ids):
        for t in (self.cur_air, self.cur_con, self.last_air, self.last_con):
            t[ids] = 0.0


def fill_state(env, g, timers, obj_timers, contact_state):
    """One step of synthetic (non-physical but well-formed) state in the B3 data-contract layout."""
    n = env.num_envs
    rd, od = env.scene["robot"].data, env.scene["object"].data
    sd, osd = env.scene.sensors["robot_contact_senosr"].data, env.scene.sensors["object_contact_sensor"].data
    rd.root_pos_w = torch.cat([U(g, (n, 2), -2, 2), U(g, (n, 1), 0.12, 0.45)], dim=1)
    rd.root_quat_w = rand_quat(g, n)
    rd.root_lin_vel_w = U(g, (n, 3), -1, 1)
    rd.root_ang_vel_w = U(g, (n, 3), -2, 2)
    rd.root_lin_vel_b = M.quat_apply_inverse(rd.root_quat_w, rd.root_lin_vel_w)
    rd.root_ang_vel_b = M.quat_apply_inverse(rd.root_quat_w, rd.root_ang_vel_w)
    rd.projected_gravity_b = M.quat_apply_inverse(rd.root_quat_w, torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1))
    rd.root_state_w = torch.cat([rd.root_pos_w, rd.root_quat_w, rd.root_lin_vel_w, rd.root_ang_vel_w], dim=1)
    rd.joint_pos = DEFAULT_Q + U(g, (n, 12), -0.9, 0.9)
    rd.joint_vel = U(g, (n, 12), -8, 8)
    rd.joint_acc = U(g, (n, 12), -300, 300)
    rd.applied_torque = U(g, (n, 12), -23.5, 23.5)
    rd.body_pos_w = torch.cat([U(g, (n, 17, 2), -2, 2), U(g, (n, 17, 1), 0.0, 0.08)], dim=2)
    rd.body_lin_vel_w = U(g, (n, 17, 3), -0.6, 0.6) * (torch.rand(n, 17, 1, generator=g) > 0.3)
    # contact forces: sparse, history newest first
    f = U(g, (n, 17, 3), -30, 30) * (torch.rand(n, 17, 1, generator=g) > 0.7)
    f = f * torch.where(torch.rand(n, 17, 1, generator=g) > 0.5, 1.0, 1.0 / 64.0)  # some tiny forces near thresholds
    prev = getattr(sd, "net_forces_w_history", torch.zeros(n, 3, 17, 3))
    sd.net_forces_w_history = torch.cat([f.unsqueeze(1), prev[:, :2]], dim=1)  # newest first
    sd.net_forces_w = f
    # Markov foot contacts at env-step rate
    flip = torch.rand(n, 17, generator=g) < 0.12
    contact_state ^= flip
    timers.update(contact_state, STEP_DT)
    sd.current_air_time, sd.current_contact_time = timers.cur_air.clone(), timers.cur_con.clone()
    sd.last_air_time, sd.last_contact_time = timers.last_air.clone(), timers.last_con.clone()
    # object
    rel = torch.cat([U(g, (n, 1), -0.2, 0.2), U(g, (n, 1), -0.15, 0.15), U(g, (n, 1), 0.05, 0.2)], dim=1)
    od.root_pos_w = rd.root_pos_w + M.quat_apply(rd.root_quat_w, rel)
    od.root_quat_w = M.quat_mul(rd.root_quat_w, M.quat_from_euler_xyz(U(g, (n,), -1.3, 1.3), U(g, (n,), -math.pi, math.pi),
                                                                       U(g, (n,), -2.0, 2.0)))
    od.root_lin_vel_w = rd.root_lin_vel_w + U(g, (n, 3), -2.5, 2.5) * (torch.rand(n, 1, generator=g) > 0.5)
    od.root_ang_vel_w = rd.root_ang_vel_w + U(g, (n, 3), -2, 2)
    od.projected_gravity_b = M.quat_apply_inverse(od.root_quat_w, torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1))
    oc = torch.rand(n, 1, generator=g) < 0.85
    obj_timers.update(oc, STEP_DT)
    osd.current_air_time, osd.current_contact_time = obj_timers.cur_air.clone(), obj_timers.cur_con.clone()
    osd.last_air_time, osd.last_contact_time = obj_timers.last_air.clone(), obj_timers.last_con.clone()
    # action term
    env.action_term.prev_raw_actions = env.action_term.raw_actions.clone()
    env.action_term.raw_actions = U(g, (n, 12), -1, 1)
    env.termination_manager.terminated = torch.rand(n, generator=g) < 0.03


GAIT_PARAMS = {  # reference locotouch/config/base/locomotion_base_env_cfg.py:166-188
    "asset_cfg": SceneEntityCfg("robot"), "sensor_cfg": SceneEntityCfg("robot_contact_senosr"),
    "synced_feet_pair_names": (("a_FR_foot", "d_RL_foot"), ("b_FL_foot", "c_RR_foot")),
    "judge_time_threshold": 1.0e-6, "air_time_gait_bound": 0.5, "contact_time_gait_bound": 0.5,
    "async_time_tolerance": 0.05, "stance_rwd_scale": 1.0, "encourage_symmetricity_and_low_frequency": 1.0,
    "soft_minimum_frequency": 2.0, "tolerance_proportion": 0.2, "rwd_upper_bound": 1.0, "rwd_lower_bound": -5.0,
    "vel_tracking_exp_sigma": 0.25, "task_performance_ratio": 1.0,
}


def resolved(name, body_names=None):
    c = SceneEntityCfg(name, body_names=body_names)
    if body_names is not None:
        ids, _ = isaaclab_shim.resolve_matching_names(body_names, BODY_NAMES if name != "object_contact_sensor" else ["Object"])
        c.body_ids = ids
    return c


def gen_rewards(n=16, T=200, seed=1234):
    """Every reward term of the RandCylinder teacher task (R1..R25) over a T-step synthetic sequence."""
    g = torch.Generator().manual_seed(seed)
    env = FakeEnv(n)
    timers, obj_timers = ContactTimers(n, 17), ContactTimers(n, 1)
    contact_state = torch.rand(n, 17, generator=g) < 0.5
    danger = dict(robot_cfg=SceneEntityCfg("robot"), object_cfg=SceneEntityCfg("object"), x_max=0.125, y_max=0.097,
                  z_min=0.095, roll_pitch_max=None, vel_xy_max=2.5)
    env.reward_cfgs["object_dangerous_state"] = RewardTermCfg(func=mdp.object_dangerous_state_ngt, weight=-50.0, params=danger)
    foot_a, foot_s = resolved("robot", ".*foot"), resolved("robot_contact_senosr", ".*foot")
    tc_s = resolved("robot_contact_senosr", [".*thigh", ".*calf"])
    obj_s = resolved("object_contact_sensor", "Object")
    # first fill so that the gait terms see valid sensor data in __init__
    fill_state(env, g, timers, obj_timers, contact_state)
    gait_cfg = RewardTermCfg(func=mdp.AdaptiveSymmetricGaitReward, weight=0.5, params=dict(GAIT_PARAMS))
    gait = mdp.AdaptiveSymmetricGaitReward(gait_cfg, env)
    gait_obj = mdp.AdaptiveSymmetricGaitRewardwithObject(gait_cfg, env)
    terms = {
        "track_lin_vel_xy": lambda: mdp.track_lin_vel_xy_pst(env, sigma=0.25, command_name="base_velocity"),
        "track_ang_vel_z": lambda: mdp.track_ang_vel_z_pst(env, sigma=0.25, command_name="base_velocity"),
        "foot_slip": lambda: mdp.foot_slipping_ngt(env, threshold=0.5, asset_cfg=foot_a, sensor_cfg=foot_s),
        "foot_dragging": lambda: mdp.foot_dragging_ngt(env, asset_cfg=foot_a, height_threshold=0.03, foot_vel_xy_threshold=0.1),
        "gait": lambda: gait(env, **GAIT_PARAMS),
        "gait_with_object": lambda: gait_obj(env, **GAIT_PARAMS),
        "track_base_height": lambda: mdp.track_base_height_ngt(env, target_height=0.42),
        "base_z_velocity": lambda: mdp.base_z_velocity_ngt(env),
        "base_roll_pitch_angle": lambda: mdp.base_roll_pitch_angle_ngt(env),
        "base_roll_pitch_velocity": lambda: mdp.base_roll_pitch_velocity_ngt(env),
        "joint_position_limit": lambda: mdp.joint_position_limit_ngt(env),
        "joint_position": lambda: mdp.joint_position_ngt(env, stand_still_scale=5.0, velocity_threshold=0.3),
        "joint_acceleration": lambda: mdp.joint_acceleration_ngt(env),
        "joint_velocity": lambda: mdp.joint_velocity_ngt(env),
        "joint_torque": lambda: mdp.joint_torque_ngt(env),
        "action_rate": lambda: mdp.action_rate_ngt(env),
        "thigh_calf_collision": lambda: mdp.thigh_calf_collision_ngt(env, threshold=0.1, sensor_cfg=tc_s),
        "object_xy_position": lambda: mdp.object_relative_xy_position_ngt(env, work_only_when_cmd=1),
        "object_xy_velocity": lambda: mdp.object_relative_xy_velocity_ngt(env),
        "object_z_contact": lambda: mdp.object_lose_contact_ngt(env, sensor_cfg=obj_s),
        "object_z_velocity": lambda: mdp.object_relative_z_velocity_ngt(env),
        "object_roll_angle": lambda: mdp.object_relative_roll_angle_ngt(env),
        "object_roll_velocity": lambda: mdp.object_relative_roll_velocity_ngt(env),
        "object_roll_pitch_angle": lambda: mdp.object_relative_roll_pitch_angle_ngt(env),
        "object_roll_pitch_velocity": lambda: mdp.object_relative_roll_pitch_velocity_ngt(env),
        "object_yaw_alignment": lambda: mdp.object_relative_yaw_angle_ngt(env, work_only_when_cmd=1),
        "object_dangerous_state": lambda: mdp.object_dangerous_state_ngt(env, **danger),
        # terminations (reference mdp/terminations.py)
        "term_object_below_robot": lambda: mdp.object_below_robot(env),
        "term_object_bad_roll": lambda: mdp.bad_roll(env, limit_angle=math.pi / 3, asset_cfg=SceneEntityCfg("object")),
        # observation (critic flavour: no noise) with the teacher cfg parameters
        "obs_object_state": lambda: mdp.object_state_in_robot_frame(
            env, sensor_cfg=obj_s, last_contact_time_threshold=1e-8, current_contact_time_threshold=1e-8,
            non_contact_obs=[0.0] * 6 + [1.0] + [0.0] * 6, add_uniform_noise=False,
            scale=[1.0] * 3 + [0.5] * 3 + [1.0] * 4 + [0.25] * 3),
    }
    IN = ["root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w", "root_lin_vel_b", "root_ang_vel_b",
          "projected_gravity_b", "joint_pos", "joint_vel", "joint_acc", "applied_torque", "body_pos_w", "body_lin_vel_w"]
    rec = {k: [] for k in
           ["cmd", "raw_actions", "prev_raw_actions", "terminated", "gait_reset", "net_forces_w",
            "current_air_time", "current_contact_time", "last_air_time", "last_contact_time",
            "obj_current_air_time", "obj_current_contact_time", "obj_last_air_time", "obj_last_contact_time",
            "obj_root_pos_w", "obj_root_quat_w", "obj_root_lin_vel_w", "obj_root_ang_vel_w", "obj_projected_gravity_b"]
           + ["robot_" + k for k in IN] + ["out_" + k for k in terms]
           + ["gait_state_" + k for k in ("valid_last_air_time", "swinging_in_zero_cmd", "valid_previous_contact",
                                          "last_velocity_cmd")]}
    cmd = torch.zeros(n, 3)
    for t in range(T):
        if t > 0:
            fill_state(env, g, timers, obj_timers, contact_state)
        # command process: piecewise constant, ~1/3 of envs at exactly zero command
        change = torch.rand(n, generator=g) < (1.0 if t == 0 else 0.03)
        new = torch.cat([U(g, (n, 1), -0.5, 0.5), U(g, (n, 1), -0.25, 0.25), U(g, (n, 1), -0.78, 0.78)], dim=1)
        new = new * (torch.rand(n, 1, generator=g) > 0.33)
        cmd = torch.where(change.unsqueeze(1), new, cmd)
        env.cmd = cmd.clone()
        # resets of the class terms (manager .reset(env_ids) before this step's call)
        rs = torch.rand(n, generator=g) < (0.0 if t == 0 else 0.01)
        ids = rs.nonzero().flatten()
        if len(ids):
            gait.reset(ids), gait_obj.reset(ids)
        rd, od = env.scene["robot"].data, env.scene["object"].data
        sd, osd = env.scene.sensors["robot_contact_senosr"].data, env.scene.sensors["object_contact_sensor"].data
        rec["cmd"].append(env.cmd.clone()), rec["gait_reset"].append(rs.clone())
        rec["raw_actions"].append(env.action_term.raw_actions.clone())
        rec["prev_raw_actions"].append(env.action_term.prev_raw_actions.clone())
        rec["terminated"].append(env.termination_manager.terminated.clone())
        rec["net_forces_w"].append(sd.net_forces_w.clone())
        for k in ("current_air_time", "current_contact_time", "last_air_time", "last_contact_time"):
            rec[k].append(getattr(sd, k)[:, 13:17].clone()), rec["obj_" + k].append(getattr(osd, k).clone())
        for k in IN:
            v = getattr(rd, k)
            rec["robot_" + k].append((v[:, 13:17] if k.startswith("body_") else v).clone())
        for k in ("root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w", "projected_gravity_b"):
            rec["obj_" + k].append(getattr(od, k).clone())
        for k, fn in terms.items():
            rec["out_" + k].append(fn().clone())
        for k in ("valid_last_air_time", "swinging_in_zero_cmd", "valid_previous_contact", "last_velocity_cmd"):
            rec["gait_state_" + k].append(getattr(gait_obj, k).clone())
    out = {k: torch.stack(v).numpy() for k, v in rec.items()}
    out["all_feet_ids"] = np.array(gait.all_feet_ids)
    out["body_names"] = np.array(BODY_NAMES)
    out["soft_joint_pos_limits"] = env.scene["robot"].data.soft_joint_pos_limits[0].numpy()
    out["default_joint_pos"] = DEFAULT_Q.numpy()
    np.savez_compressed(os.path.join(OUT, "mdp_rewards_teacher.npz"), **out)
    print("mdp_rewards_teacher.npz", {k: v.shape for k, v in out.items() if k.startswith("out_")})


def gen_actions(n=16, T=12, seed=7):
    """JointPositionActionPrevPrev (reference mdp/actions.py:13-52) over a short sequence with resets."""
    g = torch.Generator().manual_seed(seed)
    env = FakeEnv(n)
    env.scene["robot"].data.default_joint_pos = DEFAULT_Q.repeat(n, 1)
    cfg = mdp.JointPositionActionPrevPrevCfg(asset_name="robot", joint_names=[".*"], scale=1.0, use_default_offset=True,
                                             clip_raw_actions=True, raw_action_clip_value=100.0, raw_action_scale=0.25)
    term = mdp.JointPositionActionPrevPrev(cfg, env)
    rec = {k: [] for k in ("actions", "reset", "raw", "prev_raw", "prev_prev_raw", "processed", "prev_processed")}
    for t in range(T):
        a = U(g, (n, 12), -3, 3)
        a[0, 0], a[1, 1] = 500.0, -450.0  # exercise the +-100 clip
        rs = torch.rand(n, generator=g) < 0.15
        if t > 0 and rs.any():
            term.reset(rs.nonzero().flatten())
        else:
            rs = torch.zeros(n, dtype=torch.bool)
        term.process_actions(a)
        for k, v in (("actions", a), ("reset", rs), ("raw", term.raw_actions), ("prev_raw", term.prev_raw_actions),
                     ("prev_prev_raw", term.prev_prev_raw_actions), ("processed", term.processed_actions),
                     ("prev_processed", term.prev_processed_actions)):
            rec[k].append(v.clone())
    np.savez_compressed(os.path.join(OUT, "mdp_actions.npz"), **{k: torch.stack(v).numpy() for k, v in rec.items()})
    print("mdp_actions.npz")


def gen_command_zero_steps(n=32, T=80, seed=11):
    """Deterministic half of UniformVelocityCommandGaitLoggingMultiSampling (reference mdp/commands.py:561-576):
    the initial-zero-command window keyed on episode_length_buf, plus standing-env zeroing."""
    g = torch.Generator().manual_seed(seed)
    env = FakeEnv(n)
    fill = ContactTimers(n, 17)
    env.scene.sensors["robot_contact_senosr"].data.last_air_time = fill.last_air
    env.scene["robot"].data.root_lin_vel_b = torch.zeros(n, 3)
    env.scene["robot"].data.root_ang_vel_b = torch.zeros(n, 3)
    cfg = mdp.UniformVelocityCommandGaitLoggingMultiSamplingCfg(
        asset_name="robot", resampling_time_range=(8.0, 8.0), rel_heading_envs=0.0, heading_command=False,
        ranges=mdp.UniformVelocityCommandGaitLoggingMultiSamplingCfg.Ranges(
            lin_vel_x=(-0.5, 0.5), lin_vel_y=(-0.25, 0.25), ang_vel_z=(-0.78, 0.78)),
        new_command_probs=0.15, rel_standing_envs=0.05, final_rel_standing_envs=0.05,
        initial_zero_command_steps=50, final_initial_zero_command_steps=50)
    term = mdp.UniformVelocityCommandGaitLoggingMultiSampling(cfg, env)
    env.reward_cfgs["gait"] = types.SimpleNamespace(func=types.SimpleNamespace())  # no valid_last_air_time attr
    env.episode_length_buf = torch.randint(0, 45, (n,), generator=g)
    term.vel_command_b_buffer[:] = torch.cat([U(g, (n, 1), -0.5, 0.5), U(g, (n, 1), -0.25, 0.25), U(g, (n, 1), -0.78, 0.78)], 1)
    term.vel_command_b[:] = term.vel_command_b_buffer * 0.0
    term.is_standing_env[:] = torch.rand(n, generator=g) < 0.2
    rec = {k: [] for k in ("ep_len", "cmd")}
    buf, standing = term.vel_command_b_buffer.clone(), term.is_standing_env.clone()
    for t in range(T):
        env.episode_length_buf += 1
        term._update_command()
        rec["ep_len"].append(env.episode_length_buf.clone()), rec["cmd"].append(term.vel_command_b.clone())
    out = {k: torch.stack(v).numpy() for k, v in rec.items()}
    out["buffer"], out["is_standing"], out["zero_steps"] = buf.numpy(), standing.numpy(), np.array(50)
    np.savez_compressed(os.path.join(OUT, "mdp_command_zero_steps.npz"), **out)
    print("mdp_command_zero_steps.npz")


def gen_curriculum(n=48, calls=400, seed=5):
    """ModifyVelCommandsRangeBasedonReward (reference mdp/curriculums.py:184-275) + MultiSampling.set_ranges
    (mdp/commands.py:471-505): scripted reset batches -> command ranges / equal flags / zero-steps over time."""
    import contextlib
    import io

    g = torch.Generator().manual_seed(seed)
    env = FakeEnv(n)
    env.scene.sensors["robot_contact_senosr"].data.last_air_time = torch.zeros(n, 17)
    cmd_cfg = mdp.UniformVelocityCommandGaitLoggingMultiSamplingCfg(
        asset_name="robot", resampling_time_range=(8.0, 8.0), rel_heading_envs=0.0, heading_command=False,
        ranges=mdp.UniformVelocityCommandGaitLoggingMultiSamplingCfg.Ranges(
            lin_vel_x=(-0.2, 0.2), lin_vel_y=(-0.1, 0.1), ang_vel_z=(-math.pi / 10, math.pi / 10)),
        new_command_probs=0.15, rel_standing_envs=0.1, final_rel_standing_envs=0.05,
        initial_zero_command_steps=0, final_initial_zero_command_steps=50)
    env.cmd_term = mdp.UniformVelocityCommandGaitLoggingMultiSampling(cmd_cfg, env)
    env.reward_cfgs["track_lin_vel_xy"] = RewardTermCfg(func=mdp.track_lin_vel_xy_pst, weight=1.0, params={"sigma": 0.25})
    env.reward_cfgs["track_ang_vel_z"] = RewardTermCfg(func=mdp.track_ang_vel_z_pst, weight=0.5, params={"sigma": 0.25})
    params = {  # resolved RandCylinder teacher values
        "command_name": "base_velocity", "command_maximum_ranges": [0.5, 0.25, math.pi / 4],
        "curriculum_bins": [20, 20, 20], "reset_envs_episode_length": 0.98, "reward_name_lin": "track_lin_vel_xy",
        "reward_name_ang": "track_ang_vel_z", "error_threshold_lin": 0.08, "error_threshold_ang": 0.1,
        "repeat_times_lin": 1, "repeat_times_ang": 1, "max_distance_bins": 4}
    cur = mdp.ModifyVelCommandsRangeBasedonReward(CurriculumTermCfg(func=mdp.ModifyVelCommandsRangeBasedonReward, params=params), env)
    env.reward_manager._episode_sums = {"track_lin_vel_xy": torch.zeros(n), "track_ang_vel_z": torch.zeros(n)}
    rec = {k: [] for k in ("reset_mask", "ep_len", "sum_lin", "sum_ang", "ranges", "equal", "zero_steps", "rel_standing",
                           "lin_bins", "ang_bins")}
    for c in range(calls):
        # each call = one env step's reset batch; skew the stats so that gates open at different times
        mask = torch.rand(n, generator=g) < 0.2
        phase = c / calls
        env.episode_length_buf = torch.randint(5, 1001, (n,), generator=g)
        lin_ok = 1.0 if (c // 40) % 3 != 2 else 0.3
        ang_ok = 1.0 if (c // 55) % 2 == 0 else 0.3
        env.reward_manager._episode_sums["track_lin_vel_xy"] = U(g, (n,), 13.5, 19.5) * lin_ok
        env.reward_manager._episode_sums["track_ang_vel_z"] = U(g, (n,), 6.0, 9.5) * ang_ok
        ids = mask.nonzero().flatten()
        with contextlib.redirect_stdout(io.StringIO()):
            cur(env, ids, **params)
        t = env.cmd_term
        rec["reset_mask"].append(mask), rec["ep_len"].append(env.episode_length_buf.clone())
        rec["sum_lin"].append(env.reward_manager._episode_sums["track_lin_vel_xy"].clone())
        rec["sum_ang"].append(env.reward_manager._episode_sums["track_ang_vel_z"].clone())
        rec["ranges"].append(torch.tensor([t.cfg.ranges.lin_vel_x, t.cfg.ranges.lin_vel_y, t.cfg.ranges.ang_vel_z,
                                           t.cfg.previous_ranges.lin_vel_x, t.cfg.previous_ranges.lin_vel_y,
                                           t.cfg.previous_ranges.ang_vel_z], dtype=torch.float64))
        rec["equal"].append(torch.tensor([t.lin_vel_x_equal_ranges, t.lin_vel_y_equal_ranges, t.ang_vel_z_equal_ranges]))
        rec["zero_steps"].append(torch.tensor(t.initial_zero_command_steps))
        rec["rel_standing"].append(torch.tensor(t.cfg.rel_standing_envs))
        rec["lin_bins"].append(torch.tensor(cur.lin_forward_bins)), rec["ang_bins"].append(torch.tensor(cur.ang_forward_bins))
        del phase
    out = {k: torch.stack(v).numpy() for k, v in rec.items()}
    out["reward_threshold"] = np.array([cur.reward_threshold_lin, cur.reward_threshold_ang])
    out["len_threshold"] = np.array(cur.reset_envs_episode_length)
    np.savez_compressed(os.path.join(OUT, "mdp_curriculum.npz"), **out)
    print("mdp_curriculum.npz final ranges", out["ranges"][-1].tolist(), "bins", out["lin_bins"][-1], out["ang_bins"][-1])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    gen_rewards()
    gen_actions()
    gen_command_zero_steps()
    gen_curriculum()
