"""Reference env-cfg tree -> `lt_cfg` ("what cfgs hand to the env").

The unmodified launch scripts build a `ManagerBasedRLEnvCfg` tree out of the reference's own @configclass files
(`locotouch/config/**`), edit it (CLI / Hydra overrides, `*_post_init_func`s) and hand it to `gym.make(task, cfg=env_cfg)`.
The HIP env evaluates a fixed set of fused terms, so the tree is TRANSLATED here: every manager term is routed by the
identity (name) of its `func` / `class_type` to the `lt_cfg` fields that parameterise the fused implementation of exactly
that function.  Anything the kernels cannot honour - an unknown term, a different function behind a known name, an
unsupported parameter value - raises `UnsupportedCfg` instead of silently training on something else.

`translate(env_cfg)` is also how the built-in presets (`lt_cfg_preset`, locotouch_amd/csrc/lt_cfg.cpp) are pinned: a CPU
test translates the RESOLVED reference cfg of every registered LocoTouch task and requires equality with the preset
(tests/test_cfg_translate.py).
"""
from __future__ import annotations

import math

from .. import _abi

C = _abi.CONSTS


class UnsupportedCfg(NotImplementedError):
    pass


def _name(f) -> str:
    return getattr(f, "__name__", type(f).__name__)


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise UnsupportedCfg(msg)


def _close(a, b, tol=1e-9) -> bool:
    return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))


def _terms(section) -> dict:
    return {k: v for k, v in vars(section).items() if not k.startswith("_") and v is not None}


def _rng(dst, value) -> None:
    dst[0], dst[1] = float(value[0]), float(value[1])


# reward term name -> (enum, {allowed func names}, param routing)
_REWARDS = {
    "alive": ("LT_R_ALIVE", {"is_alive"}),
    "track_lin_vel_xy": ("LT_R_TRACK_LIN_VEL_XY", {"track_lin_vel_xy_pst"}),
    "track_ang_vel_z": ("LT_R_TRACK_ANG_VEL_Z", {"track_ang_vel_z_pst"}),
    "foot_slip": ("LT_R_FOOT_SLIP", {"foot_slipping_ngt"}),
    "foot_dragging": ("LT_R_FOOT_DRAGGING", {"foot_dragging_ngt"}),
    "gait": ("LT_R_GAIT", {"AdaptiveSymmetricGaitReward", "AdaptiveSymmetricGaitRewardwithObject"}),
    "track_base_height": ("LT_R_TRACK_BASE_HEIGHT", {"track_base_height_ngt"}),
    "base_z_velocity": ("LT_R_BASE_Z_VELOCITY", {"base_z_velocity_ngt"}),
    "base_roll_pitch_angle": ("LT_R_BASE_ROLL_PITCH_ANGLE", {"base_roll_pitch_angle_ngt"}),
    "base_roll_pitch_velocity": ("LT_R_BASE_ROLL_PITCH_VELOCITY", {"base_roll_pitch_velocity_ngt"}),
    "joint_position_limit": ("LT_R_JOINT_POSITION_LIMIT", {"joint_position_limit_ngt"}),
    "joint_position": ("LT_R_JOINT_POSITION", {"joint_position_ngt"}),
    "joint_acceleration": ("LT_R_JOINT_ACCELERATION", {"joint_acceleration_ngt"}),
    "joint_velocity": ("LT_R_JOINT_VELOCITY", {"joint_velocity_ngt"}),
    "joint_torque": ("LT_R_JOINT_TORQUE", {"joint_torque_ngt"}),
    "action_rate": ("LT_R_ACTION_RATE", {"action_rate_ngt"}),
    "thigh_calf_collision": ("LT_R_THIGH_CALF_COLLISION", {"thigh_calf_collision_ngt"}),
    "object_xy_position": ("LT_R_OBJECT_XY_POSITION", {"object_relative_xy_position_ngt"}),
    "object_xy_velocity": ("LT_R_OBJECT_XY_VELOCITY", {"object_relative_xy_velocity_ngt"}),
    "object_z_contact": ("LT_R_OBJECT_Z_CONTACT", {"object_lose_contact_ngt"}),
    "object_z_velocity": ("LT_R_OBJECT_Z_VELOCITY", {"object_relative_z_velocity_ngt"}),
    # the fused kernel implements the cylinder variants (roll only); the roll+pitch variants of the box teacher are accepted
    # at weight 0 only (they are never evaluated then)
    "object_roll_pitch_angle": ("LT_R_OBJECT_ROLL_PITCH_ANGLE", {"object_relative_roll_angle_ngt"}),
    "object_roll_pitch_velocity": ("LT_R_OBJECT_ROLL_PITCH_VELOCITY", {"object_relative_roll_velocity_ngt"}),
    "object_yaw_alignment": ("LT_R_OBJECT_YAW_ALIGNMENT", {"object_relative_yaw_angle_ngt"}),
    "object_dangerous_state": ("LT_R_OBJECT_DANGEROUS_STATE", {"object_dangerous_state_ngt"}),
}
_TERMINATIONS = {
    "time_out": ("LT_T_TIME_OUT", "time_out"),
    "base_orientation": ("LT_T_BASE_ORIENTATION", "bad_orientation"),
    "base_height_below_minimum": ("LT_T_BASE_HEIGHT", "root_height_below_minimum"),
    "base_contact": ("LT_T_BASE_CONTACT", "illegal_contact"),
    "hip_contact": ("LT_T_HIP_CONTACT", "illegal_contact"),
    "object_below_robot": ("LT_T_OBJECT_BELOW_ROBOT", "object_below_robot"),
    "object_bad_orientation": ("LT_T_OBJECT_BAD_ROLL", "bad_roll"),
}
_POSE_KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


def task_kind(env_cfg) -> int:
    """LT_TASK_* of a cfg tree: the transport tasks have an `object` in the scene."""
    return C["LT_TASK_TRANSPORT_TEACHER"] if getattr(env_cfg.scene, "object", None) is not None else C["LT_TASK_LOCOMOTION"]


# The student -Play- registration adds two 4-channel tactile groups next to `tactile` (object_transport_student_env_cfg.py:166-171;
# read by the ROS publisher block of distillation.py:190-199, which the reference keeps switched off).  They are computed
# (cfg.tactile_aux_groups); `translate(..., omit_groups=VISUALISATION_ONLY_GROUPS)` still lets a caller leave them out by name.
VISUALISATION_ONLY_GROUPS = ("original_tactile", "processed_tactile")
TACTILE_FORMATS = {"BinaryTactileSignals": "LT_TACTILE_BINARY", "NormalizedTactileSignals": "LT_TACTILE_NORMALIZED",
                   "DiscreteTactileSignals": "LT_TACTILE_DISCRETE", "CotinuousTactileSignals": "LT_TACTILE_CONTINUOUS",
                   "ProcessedTactileSignals": "LT_TACTILE_PROCESSED", "TactileSignals": "LT_TACTILE_ORIGINAL"}


def translate(env_cfg, seed: int | None = None, omit_groups: tuple = (), collect_unknown_rewards: bool = False) -> "_abi.LtCfg":
    """`collect_unknown_rewards`: a reward term the fused kernels do not know is not an error - it is returned in
    `cfg.extra_reward_terms` [(name, func, weight, params)], an unknown termination term in
    `cfg.extra_termination_terms` [(name, func, params, time_out)], for the slow torch path (compat/scene_views.py: evaluated on
    IsaacLab-layout views after every step and added to the kernel's reward)."""
    kind = task_kind(env_cfg)
    has_obj = kind != C["LT_TASK_LOCOMOTION"]
    cfg = _abi.default_cfg(kind, num_envs=int(env_cfg.scene.num_envs), seed=int(seed if seed is not None else (getattr(env_cfg, "seed", None) or 42)))
    # everything a term can switch off starts switched off; the tree switches things on
    for i in range(len(cfg.reward_weight)):
        cfg.reward_weight[i] = 0.0
    for i in range(len(cfg.term_enabled)):
        cfg.term_enabled[i] = 0
    cfg.cur_enabled = 0

    # ---- timing (locomotion_base_env_cfg.py:343-349) ----
    cfg.sim_dt = float(env_cfg.sim.dt)
    cfg.decimation = int(env_cfg.decimation)
    cfg.episode_length_s = float(env_cfg.episode_length_s)
    cfg.max_episode_length = int(math.ceil(cfg.episode_length_s / (float(env_cfg.sim.dt) * cfg.decimation)))
    pm = getattr(env_cfg.sim, "physics_material", None)
    if pm is not None:
        _need(getattr(pm, "friction_combine_mode", "multiply") == "multiply", "ground friction combine mode must be 'multiply'")
        cfg.ground_mu = float(min(pm.static_friction, pm.dynamic_friction))

    # ---- robot: DC-motor actuator (assets/go1.py:41-49) ----
    acts = getattr(env_cfg.scene.robot, "actuators", None) or {}
    _need(len(acts) == 1, "exactly one actuator group (Go1 legs) is supported")
    a = next(iter(acts.values()))
    _need(type(a).__name__ == "DCMotorCfg", f"actuator model {type(a).__name__} is not the DC motor of the Go1 asset")
    cfg.kp, cfg.kd = float(a.stiffness), float(a.damping)
    cfg.effort_limit, cfg.saturation_effort, cfg.velocity_limit = float(a.effort_limit), float(a.saturation_effort), float(a.velocity_limit)
    init = env_cfg.scene.robot.init_state
    _need(_close(init.pos[2], 0.28, 1e-6), "robot init height differs from the compiled Go1 model (0.28 m)")

    # ---- action term (mdp/actions.py:30-44) ----
    at = _terms(env_cfg.actions)
    _need(set(at) == {"joint_pos"}, f"action terms {sorted(at)}: only `joint_pos` is supported")
    ja = at["joint_pos"]
    _need(_name(ja.class_type) == "JointPositionActionPrevPrev", f"action class {_name(ja.class_type)} unsupported")
    _need(_close(ja.scale, 1.0) and ja.use_default_offset and ja.clip is None and list(ja.joint_names) == [".*"],
          "joint_pos action: scale 1.0, default offset, all joints, no processed-action clip")
    cfg.action_clip = float(ja.raw_action_clip_value) if ja.clip_raw_actions else 3.0e38
    cfg.action_scale = float(ja.raw_action_scale)

    # ---- command term (mdp/commands.py:379-576) ----
    ct = _terms(env_cfg.commands)
    _need(set(ct) == {"base_velocity"}, "exactly one command term `base_velocity`")
    cc = ct["base_velocity"]
    cname = _name(cc.class_type)
    _need(cname in ("UniformVelocityCommandGaitLogging", "UniformVelocityCommandGaitLoggingMultiSampling"), f"command class {cname} unsupported")
    _need(not cc.heading_command, "heading commands are not implemented")
    cfg.cmd_multi_sampling = 1 if cname.endswith("MultiSampling") else 0
    cfg.cmd_binary_maximal = 1 if getattr(cc, "binary_maximal_command", False) else 0  # (commands.py:95-104, 452-461: both classes)
    for d, key in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_z")):
        _rng(cfg.cmd_range_init[d], getattr(cc.ranges, key))
        cfg.cmd_range_max[d] = float(getattr(cc.ranges, key)[1])
    _rng(cfg.cmd_resample_time, cc.resampling_time_range)
    cfg.cmd_rel_standing = cfg.cmd_rel_standing_final = float(cc.rel_standing_envs)
    cfg.cmd_zero_steps = cfg.cmd_zero_steps_final = 0
    cfg.cmd_new_probs = 0.15
    if cfg.cmd_multi_sampling:
        cfg.cmd_new_probs = float(cc.new_command_probs)
        cfg.cmd_rel_standing_final = float(cc.final_rel_standing_envs)
        cfg.cmd_zero_steps = int(cc.initial_zero_command_steps)
        cfg.cmd_zero_steps_final = int(cc.final_initial_zero_command_steps)

    # ---- rewards (mdp/rewards.py; RewardManager: weight * dt, zero weight = skipped) ----
    rt = _terms(env_cfg.rewards)
    sigma = {}
    extra_terms = []
    def to_slow_path(name, term, why):
        """A term under a fused NAME that the fused implementation cannot honour (another function, a parameter value the kernel
        does not implement): with `collect_unknown_rewards` it is evaluated by the slow path with the cfg's own function -
        locotouch/mdp/rewards.py:469-481,503-522,545-594 run as they are on the views of compat/scene_views.py - and its fused
        weight stays 0; strict translation refuses."""
        _need(collect_unknown_rewards, f"reward {name!r}: {why}")
        if float(term.weight) != 0.0:
            extra_terms.append((name, term.func, float(term.weight), dict(term.params or {})))

    for name, term in rt.items():
        if name not in _REWARDS and collect_unknown_rewards:
            if float(term.weight) != 0.0:
                extra_terms.append((name, term.func, float(term.weight), dict(term.params or {})))
            continue
        _need(name in _REWARDS, f"reward term {name!r} has no fused implementation")
        enum, funcs = _REWARDS[name]
        w = float(term.weight)
        fn = _name(term.func)
        if fn not in funcs:
            if w != 0.0:
                to_slow_path(name, term, f"func {fn} is not the implemented {sorted(funcs)}")
            continue
        p = term.params
        if name in ("object_xy_position", "object_yaw_alignment") and int(p.get("work_only_when_cmd", 0)) != 1 and w != 0.0:
            to_slow_path(name, term, "work_only_when_cmd must be 1")  # (the function's default is 0, rewards.py:473,549)
            continue
        if name == "object_dangerous_state":
            # x_max / y_max are also read by the gait-with-object class (rewards.py:380-381), whichever path evaluates the term
            cfg.danger_x_max, cfg.danger_y_max = float(p["x_max"]), float(p["y_max"])
            if p.get("roll_pitch_max") is not None or any(p.get(k) is None for k in ("z_min", "vel_xy_max")):
                if w != 0.0:
                    to_slow_path(name, term, "roll_pitch_max / a disabled limit is not implemented by the fused term")
                continue
            cfg.danger_z_min, cfg.danger_vel_xy_max = float(p["z_min"]), float(p["vel_xy_max"])
        cfg.reward_weight[C[enum]] = w
        if name in ("track_lin_vel_xy", "track_ang_vel_z"):
            sigma[name] = float(p["sigma"])
        elif name == "foot_slip":
            cfg.foot_slip_threshold = float(p["threshold"])
        elif name == "foot_dragging":
            cfg.foot_drag_height, cfg.foot_drag_vel = float(p["height_threshold"]), float(p["foot_vel_xy_threshold"])
        elif name == "track_base_height":
            cfg.base_height_target = float(p["target_height"])
        elif name == "joint_position":
            cfg.joint_pos_stand_scale, cfg.joint_pos_vel_threshold = float(p["stand_still_scale"]), float(p["velocity_threshold"])
        elif name == "thigh_calf_collision":
            cfg.thigh_calf_threshold = float(p["threshold"])
        elif name == "gait":
            cfg.gait_with_object = 1 if fn.endswith("withObject") else 0
            _need(tuple(map(tuple, p["synced_feet_pair_names"])) == (("a_FR_foot", "d_RL_foot"), ("b_FL_foot", "c_RR_foot")), "gait: foot pairs differ from the trot pairs")
            cfg.gait_judge_time = float(p["judge_time_threshold"])
            cfg.gait_air_bound, cfg.gait_contact_bound = float(p["air_time_gait_bound"]), float(p["contact_time_gait_bound"])
            cfg.gait_async_tolerance = float(p["async_time_tolerance"])
            cfg.gait_stance_scale = float(p.get("stance_rwd_scale", 1.0))
            cfg.gait_soft_min_frequency = float(p["encourage_symmetricity_and_low_frequency"]) if not isinstance(p.get("encourage_symmetricity_and_low_frequency"), (bool, type(None))) else cfg.gait_soft_min_frequency
            for key, field in (("soft_minimum_frequency", "gait_soft_min_frequency"), ("tolerance_proportion", "gait_tolerance_proportion"),
                               ("rwd_upper_bound", "gait_rwd_upper"), ("rwd_lower_bound", "gait_rwd_lower"),
                               ("vel_tracking_exp_sigma", "gait_vel_sigma"), ("task_performance_ratio", "gait_task_ratio")):
                if key in p:
                    setattr(cfg, field, float(p[key]))
    if "track_lin_vel_xy" in sigma and "track_ang_vel_z" in sigma:
        _need(_close(sigma["track_lin_vel_xy"], sigma["track_ang_vel_z"]), "the two tracking rewards must share sigma")
    if sigma:
        cfg.track_sigma = next(iter(sigma.values()))
    if cfg.gait_with_object:
        _need(has_obj, "gait-with-object needs an object in the scene")

    # ---- terminations (locomotion_base_env_cfg.py:296-313, mdp/terminations.py) ----
    extra_terminations = []
    for name, term in _terms(env_cfg.terminations).items():
        if name not in _TERMINATIONS and collect_unknown_rewards:
            # slow path (compat/scene_views.py): LT_T_USER, or LT_T_USER_TIME_OUT for a term flagged time_out = True
            extra_terminations.append((name, term.func, dict(term.params or {}), bool(term.time_out)))
            continue
        _need(name in _TERMINATIONS, f"termination term {name!r} has no fused implementation")
        enum, fn = _TERMINATIONS[name]
        _need(_name(term.func) == fn, f"termination {name!r}: func {_name(term.func)} != {fn}")
        _need(bool(term.time_out) == (name == "time_out"), f"termination {name!r}: unexpected time_out flag")
        cfg.term_enabled[C[enum]] = 1
        p = term.params
        if name == "base_orientation":
            cfg.term_orientation_limit = float(p["limit_angle"])
        elif name == "base_height_below_minimum":
            cfg.term_min_height = float(p["minimum_height"])
        elif name in ("base_contact", "hip_contact"):
            exp = ".*hip" if name == "hip_contact" else "trunk"
            _need(p["sensor_cfg"].body_names == exp, f"{name}: body_names {p['sensor_cfg'].body_names!r} != {exp!r}")
            cfg.term_contact_threshold = float(p["threshold"])
        elif name == "object_bad_orientation":
            cfg.term_object_roll_limit = float(p["limit_angle"])

    # ---- observations (policy group: noisy; critic: same terms, corruption off) ----
    groups = _terms(env_cfg.observations)
    omitted = sorted(set(groups) & set(omit_groups))
    if omitted:
        import warnings

        warnings.warn(f"observation groups {omitted} left out on request (visualisation-only in the reference)")
        groups = {k: v for k, v in groups.items() if k not in omitted}
    extra = set(groups) - {"policy", "critic", "tactile", "object_state", "original_tactile", "processed_tactile"}
    _need(not extra, f"observation groups {sorted(extra)} have no fused implementation")
    _need("tactile" in groups or not (set(groups) & {"original_tactile", "processed_tactile"}), "the 4-channel tactile groups come with the `tactile` group")
    _need(("tactile" in groups) == ("object_state" in groups), "the student tasks carry the tactile and object_state groups together")
    pol = groups["policy"]
    cfg.enable_corruption = 1 if pol.enable_corruption else 0
    hist = {int(v.history_length) for v in vars(pol).values() if hasattr(v, "func") and v is not None and v.history_length is not None}
    if pol.history_length is not None:
        hist = {int(pol.history_length)}  # a group-level history overrides the terms' (ObservationManager [DEP])
    _need(len(hist) == 1, f"observation terms must share one history length, got {sorted(hist)}")
    cfg.obs_history = hist.pop()
    order = ["velocity_commands", "base_ang_vel", "projected_gravity", "joint_pos", "joint_vel", "last_action"] + (["object_state"] if has_obj else [])
    pterms = {k: v for k, v in vars(pol).items() if hasattr(v, "func") and v is not None}
    # terms BEHIND the fused ones (a derived cfg class appends its attributes): served by the slow path (compat/scene_views.py
    # ExtraTerms.add_observation) when the caller collects unknown terms; their columns follow the kernel's rows
    user_obs = []

    def collect_user_terms(group_name, group, terms):
        _need(collect_unknown_rewards, f"{group_name} observation terms {list(terms)[len(order):]} have no fused implementation")
        for k in list(terms)[len(order):]:
            t = terms[k]
            _need(not getattr(t, "modifiers", None), f"observation {k!r}: modifiers are not implemented")
            noise = None
            if t.noise is not None and group.enable_corruption:
                _need(hasattr(t.noise, "n_min") and getattr(t.noise, "operation", "add") == "add", f"observation {k!r}: only additive uniform noise")
                noise = (float(t.noise.n_min), float(t.noise.n_max))
            h = group.history_length if group.history_length is not None else t.history_length
            user_obs.append(dict(group=group_name, name=k, func=t.func, params=dict(t.params or {}), history_length=int(h or 0),
                                 scale=t.scale, clip=t.clip, noise=noise))

    if list(pterms)[:len(order)] == order and len(pterms) > len(order):
        collect_user_terms("policy", pol, pterms)
        pterms = {k: pterms[k] for k in order}
    _need(list(pterms) == order, f"policy observation terms {list(pterms)} != {order}")
    funcs = ["generated_commands", "base_ang_vel", "projected_gravity", "joint_pos_rel", "joint_vel_rel", "last_action", "object_state_in_robot_frame"]
    for (k, t), fn in zip(pterms.items(), funcs):
        _need(_name(t.func) == fn, f"observation {k!r}: func {_name(t.func)} != {fn}")
        _need(t.clip is None and not t.modifiers, f"observation {k!r}: clip / modifiers are not implemented")

    def sym(noise):
        if noise is None:
            return 0.0
        _need(_close(-noise.n_min, noise.n_max) and getattr(noise, "operation", "add") == "add", "observation noise must be symmetric additive uniform")
        return float(noise.n_max)

    for k in ("velocity_commands", "last_action"):
        _need(pterms[k].noise is None and _close(pterms[k].scale, 1.0), f"observation {k!r}: no noise, scale 1")
    _need(_close(pterms["projected_gravity"].scale, 1.0) and _close(pterms["joint_pos"].scale, 1.0), "gravity / joint_pos observation scale must be 1")
    cfg.obs_noise_ang_vel, cfg.obs_scale_ang_vel = sym(pterms["base_ang_vel"].noise), float(pterms["base_ang_vel"].scale)
    cfg.obs_noise_gravity = sym(pterms["projected_gravity"].noise)
    cfg.obs_noise_joint_pos = sym(pterms["joint_pos"].noise)
    cfg.obs_noise_joint_vel, cfg.obs_scale_joint_vel = sym(pterms["joint_vel"].noise), float(pterms["joint_vel"].scale)
    if has_obj:
        op = pterms["object_state"].params
        _need(_close(pterms["object_state"].scale, 1.0) and pterms["object_state"].noise is None, "object_state: manager-level noise/scale unused by the reference")
        _need(list(op["non_contact_obs"]) == [0.0] * 6 + [1.0] + [0.0] * 6, "object_state: non_contact_obs")
        _need(_close(op["last_contact_time_threshold"], op["current_contact_time_threshold"]), "object_state: one contact-time threshold")
        cfg.obj_contact_time_threshold = float(op["last_contact_time_threshold"])
        nmin, nmax = list(op["n_min"]), list(op["n_max"])
        _need(len(nmin) == 12 and all(_close(-a_, b_) for a_, b_ in zip(nmin, nmax)), "object_state noise: 12 symmetric half-widths")
        for i in range(12):
            cfg.obj_noise[i] = float(nmax[i]) if op["add_uniform_noise"] else 0.0
        cfg.obj_noise[12] = 0.0
        sc = op["scale"]
        for i in range(13):
            cfg.obj_scale[i] = float(sc[i]) if not isinstance(sc, float) else float(sc)
    if "tactile" in groups:
        _translate_student_groups(env_cfg, cfg, groups, pterms)
    cri = groups["critic"]
    _need(not cri.enable_corruption, "critic group: corruption off")
    cterms = {k: v for k, v in vars(cri).items() if hasattr(v, "func") and v is not None}
    if list(cterms)[:len(order)] == order and len(cterms) > len(order):
        collect_user_terms("critic", cri, cterms)
        cterms = {k: cterms[k] for k in order}
    _need(list(cterms) == order, "critic group must hold the policy group's terms")
    cfg.extra_observation_terms = user_obs

    # ---- events ----
    seen = set()
    for name, ev in _terms(env_cfg.events).items():
        fn, p = _name(ev.func), ev.params
        ac = p.get("asset_cfg")
        asset = getattr(ac, "name", "robot")
        bodies = getattr(ac, "body_names", None)
        key = (fn, asset, ev.mode)
        _need(key not in seen, f"event {name!r}: a second {key} term")
        seen.add(key)
        if key == ("randomize_rigid_body_mass", "robot", "startup"):
            _need(bodies == "trunk" and p["operation"] == "add", "robot mass randomisation: trunk, additive")
            _rng(cfg.trunk_mass_add, p["mass_distribution_params"])
        elif key == ("randomize_rigid_body_mass", "object", "reset"):
            _need(p["operation"] == "add", "object mass randomisation: additive")
            _rng(cfg.obj_mass_add, p["mass_distribution_params"])
        elif key == ("randomize_rigid_body_material", "robot", "startup"):
            _need(bodies == ".*foot" and p["make_consistent"] and tuple(p["static_friction_range"]) == tuple(p["dynamic_friction_range"]),
                  "foot material: feet, consistent, one friction range")
            _rng(cfg.foot_friction, p["static_friction_range"])
            _rng(cfg.foot_restitution, p["restitution_range"])
            cfg.foot_material_buckets = int(p["num_buckets"])
        elif key == ("randomize_rigid_body_material", "object", "reset"):
            _need(p["make_consistent"] and tuple(p["dynamic_friction_range"]) == (1.0, 1.0), "object material: consistent, dynamic range (1, 1)")
            _rng(cfg.obj_friction, p["static_friction_range"])
            _rng(cfg.obj_restitution, p["restitution_range"])
            cfg.obj_material_buckets = int(p["num_buckets"])
        elif key == ("randomize_friction_restitution", "robot", "reset"):
            _need(bodies == "trunk" and p["make_consistent"] and tuple(p["dynamic_friction_range"]) == (1.0, 1.0), "trunk material: consistent, dynamic range (1, 1)")
            _rng(cfg.trunk_friction, p["static_friction_range"])
            _rng(cfg.trunk_restitution, p["restitution_range"])
        elif key == ("reset_root_state_uniform", "robot", "reset"):
            for i, k in enumerate(_POSE_KEYS):
                v = p["pose_range"].get(k, (0.0, 0.0))
                _rng(cfg.reset_root_pos[i] if i < 3 else cfg.reset_root_rpy[i - 3], v)
                _rng(cfg.reset_root_vel[i], p["velocity_range"].get(k, (0.0, 0.0)))
        elif key == ("reset_joints_by_offset", "robot", "reset"):
            _rng(cfg.reset_joint_pos, p["position_range"])
            _rng(cfg.reset_joint_vel, p["velocity_range"])
        elif key == ("ResetObjectStateUniform", "object", "reset"):
            _need(not any(tuple(v) != (0.0, 0.0) for v in p.get("velocity_range", {}).values()), "object reset: velocity offsets are not implemented")
            cfg.obj_reset_robot_frame = 0
            for i, k in enumerate(_POSE_KEYS):
                _rng(cfg.obj_reset_pos[i] if i < 3 else cfg.obj_reset_rpy[i - 3], p["pose_range"].get(k, (0.0, 0.0)))
        elif key == ("reset_object_state_uniform", "object", "reset"):  # function variant: offset in the robot's axes (events.py:13-53)
            _need(not any(tuple(v) != (0.0, 0.0) for v in p.get("velocity_range", {}).values()), "object reset: velocity offsets are not implemented")
            cfg.obj_reset_robot_frame = 1
            for i, k in enumerate(_POSE_KEYS):
                _rng(cfg.obj_reset_pos[i] if i < 3 else cfg.obj_reset_rpy[i - 3], p["pose_range"].get(k, (0.0, 0.0)))
        elif key == ("push_by_setting_velocity", "robot", "interval"):
            _rng(cfg.push_robot_interval, ev.interval_range_s)
            for i, k in enumerate(_POSE_KEYS):
                _rng(cfg.push_robot_vel[i], p["velocity_range"].get(k, (0.0, 0.0)))
        elif key == ("push_by_setting_velocity", "object", "interval"):
            _rng(cfg.push_obj_interval, ev.interval_range_s)
            for i, k in enumerate(_POSE_KEYS):
                _rng(cfg.push_obj_vel[i], p["velocity_range"].get(k, (0.0, 0.0)))
        else:
            raise UnsupportedCfg(f"event term {name!r} ({fn} on {asset!r}, mode {ev.mode!r}) has no fused implementation")
    if ("push_by_setting_velocity", "robot", "interval") not in seen:
        _rng(cfg.push_robot_interval, (1.0e9, 1.0e9))
    if has_obj and ("push_by_setting_velocity", "object", "interval") not in seen:
        _rng(cfg.push_obj_interval, (1.0e9, 1.0e9))

    # ---- object (cylinder) ----
    if has_obj:
        sp = env_cfg.scene.object.spawn
        kind_name = type(sp).__name__
        if kind_name == "CylinderCfg":
            _need(sp.axis == "Y", "cylinder axis must be Y")
            _rng(cfg.obj_radius, (sp.radius, sp.radius))
            _rng(cfg.obj_length, (sp.height, sp.height))
            mass = sp.mass_props.mass
        elif kind_name == "MultiAssetSpawnerCfg":
            cyl = sp.assets_cfg
            _need(all(type(c_).__name__ == "CylinderCfg" and c_.axis == "Y" for c_ in cyl), "multi-asset object: Y-axis cylinders only")
            # per-env sizes are drawn (seeded, quirk Q2) inside the env from the range the cfg's samples span
            _rng(cfg.obj_radius, (min(c_.radius for c_ in cyl), max(c_.radius for c_ in cyl)))
            _rng(cfg.obj_length, (min(c_.height for c_ in cyl), max(c_.height for c_ in cyl)))
            mass = sp.mass_props.mass
        else:
            raise UnsupportedCfg(f"object spawner {kind_name} is not a cylinder")
        _need(_close(mass, 1.0), "object base mass must be 1.0 kg")

    # ---- curriculum (mdp/curriculums.py:184-275) ----
    cur = _terms(env_cfg.curriculum) if getattr(env_cfg, "curriculum", None) is not None else {}
    if cur:
        _need(set(cur) == {"velocity_commands"} and _name(cur["velocity_commands"].func) == "ModifyVelCommandsRangeBasedonReward",
              f"curriculum terms {sorted(cur)}: only ModifyVelCommandsRangeBasedonReward is implemented")
        _need(cfg.cmd_multi_sampling == 1, "the velocity curriculum needs the MultiSampling command term")
        p = cur["velocity_commands"].params
        cfg.cur_enabled = 1
        for d in range(3):
            cfg.cmd_range_max[d] = float(p["command_maximum_ranges"][d])
            cfg.cur_bins[d] = int(p["curriculum_bins"][d])
        cfg.cur_len_threshold = float(p["reset_envs_episode_length"]) * cfg.episode_length_s       # :194 (quirk Q4)
        lin, ang = rt[p["reward_name_lin"]], rt[p["reward_name_ang"]]
        cfg.cur_reward_threshold[0] = math.exp(-float(p["error_threshold_lin"]) / float(lin.params["sigma"])) * float(lin.weight) * cfg.episode_length_s  # :199
        cfg.cur_reward_threshold[1] = math.exp(-float(p["error_threshold_ang"]) / float(ang.params["sigma"])) * float(ang.weight) * cfg.episode_length_s  # :200
        cfg.cur_repeat_times[0], cfg.cur_repeat_times[1] = int(p["repeat_times_lin"]), int(p["repeat_times_ang"])
        cfg.cur_max_distance_bins = int(p["max_distance_bins"])
    cfg.extra_reward_terms = extra_terms  # (a Python attribute beside the C struct: empty unless collect_unknown_rewards)
    cfg.extra_termination_terms = extra_terminations
    return cfg


def _translate_student_groups(env_cfg, cfg, groups, pterms) -> None:
    """`tactile` (any TactileSignals class on the 17 x 13 taxel sensor), the -Play- env's 4-channel groups and `object_state`
    (reference config/locotouch/object_transport_student_env_cfg.py:13-43,161-201)."""
    _need(cfg.task == C["LT_TASK_TRANSPORT_TEACHER"], "tactile observations need the transport scene")
    sensor = getattr(env_cfg.scene, "tactile_contact_sensor", None)
    _need(sensor is not None and sensor.prim_path.endswith("/Robot/sensor_.*"), "scene.tactile_contact_sensor on the taxel bodies")
    cfg.tactile_enabled = 1
    cfg.tactile_update_period = float(sensor.update_period)

    def term_params(gname, expect_func=None):
        """(format, parameter tuple) of a tactile group's single term; every TactileSignals class is served (observations.py:248-429)."""
        grp = groups[gname]
        tterms = {k: v for k, v in vars(grp).items() if hasattr(v, "func") and v is not None}
        _need(list(tterms) == ["tactile_signals"], f"{gname} group terms {list(tterms)} != ['tactile_signals']")
        t = tterms["tactile_signals"]
        _need(_name(t.func) in TACTILE_FORMATS, f"{gname} term {_name(t.func)} is not a TactileSignals class")
        _need(expect_func is None or _name(t.func) == expect_func, f"{gname} group must be {expect_func}, got {_name(t.func)}")
        _need(grp.enable_corruption and grp.concatenate_terms and not t.history_length and t.noise is None and _close(t.scale, 1.0)
              and t.clip is None and not t.modifiers, f"{gname} group: corruption on, concatenated, no history / manager noise / scale / clip")
        p = t.params
        _need(tuple(p["tactile_signal_shape"]) == (C["LT_TACTILE_ROWS"], C["LT_TACTILE_COLS"]), "taxel grid must be 17 x 13")
        _need(not float(p.get("add_continuous_artifact", 0.0)) > 0.5, "continuous tactile artifacts are not implemented (nor by the reference: the flag is read and never used)")

        def half(flag, lo, hi, what):
            if not p[flag]:
                return 0.0
            _need(_close(-float(p[lo]), float(p[hi])), f"tactile {what} noise must be symmetric")
            return float(p[hi])

        return C[TACTILE_FORMATS[_name(t.func)]], (float(p["contact_threshold"]), half("add_threshold_noise", "threshold_n_min", "threshold_n_max", "threshold"),
                                                   float(p["contact_dropout_prob"]), float(p["contact_addition_prob"]),
                                                   half("add_force_noise", "force_n_prop_min", "force_n_prop_max", "force"), float(p["maximal_force"]),
                                                   int(p["total_levels"]), half("add_level_noise", "level_n_min", "level_n_max", "level"))

    fmt, par = term_params("tactile")
    cfg.tactile_format = fmt
    (cfg.tactile_threshold, cfg.tactile_threshold_noise, cfg.tactile_dropout_prob, cfg.tactile_addition_prob, cfg.tactile_force_noise,
     cfg.tactile_maximal_force, cfg.tactile_total_levels, cfg.tactile_level_noise) = par
    cfg.tactile_aux_groups = 0
    for bit, gname, func in ((1, "original_tactile", "TactileSignals"), (2, "processed_tactile", "ProcessedTactileSignals")):
        if gname in groups:
            _, par_aux = term_params(gname, func)
            _need(all(_close(a_, b_) for a_, b_ in zip(par_aux, par)), f"{gname}: the tactile groups of an env share one parameter set (lt_cfg.tactile_*)")
            cfg.tactile_aux_groups |= bit
    # object_state group: served as a window of the policy rows, so it must be the policy group's own object_state term
    og = groups["object_state"]
    oterms = {k: v for k, v in vars(og).items() if hasattr(v, "func") and v is not None}
    _need(list(oterms) == ["object_state"] and og.enable_corruption and og.concatenate_terms, "object_state group: one noisy term")
    a, b = oterms["object_state"], pterms["object_state"]
    _need(_name(a.func) == _name(b.func) and int(a.history_length or 0) == int(cfg.obs_history)
          and all(_same(a.params.get(k), b.params.get(k)) for k in set(a.params) | set(b.params) if not k.endswith("_cfg")),
          "object_state group must repeat the policy group's object_state term (it is served as a window of the policy rows)")


def _same(x, y) -> bool:
    if isinstance(x, (list, tuple)) and isinstance(y, (list, tuple)):
        return len(x) == len(y) and all(_same(a, b) for a, b in zip(x, y))
    if isinstance(x, (int, float)) and isinstance(y, (int, float)):
        return _close(x, y)
    return x == y


def diff(a: "_abi.LtCfg", b: "_abi.LtCfg", rtol: float = 1e-6, skip=("seed", "num_envs", "reserved", "debug_terms")) -> list[str]:
    """Field-by-field differences of two lt_cfg values (flattened arrays), for tests and for loud warnings."""
    import numpy as np

    out = []
    for name, _ in a._fields_:
        if name in skip:
            continue
        va, vb = np.array(getattr(a, name), dtype=np.float64).reshape(-1), np.array(getattr(b, name), dtype=np.float64).reshape(-1)
        if not np.allclose(va, vb, rtol=rtol, atol=1e-9):
            out.append(f"{name}: {va.tolist()} != {vb.tolist()}")
    return out
