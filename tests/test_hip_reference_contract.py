"""GPU twin of tests/test_reference_terms_slow_path.py: terms written against the REFERENCE'S data contract -
`SceneEntityCfg("robot_contact_senosr", body_names=...)`, `asset.data.body_lin_vel_w[:, body_ids]`,
`action_manager.get_term("joint_pos").raw_actions`, `object_contact_sensor` timers, `reward_manager.get_term_cfg(...)` - run on the
views over the HIP env's arena and must equal the fused kernel's own columns of LT_F_REWARD_TERMS.  The term bodies below are
restated locally (what they read and compute follows locotouch/mdp/rewards.py:31-56,454-466,569-604); the reference checkout does
not exist on the GPU box.  Also: the user TIME-OUT hook (LT_T_USER_TIME_OUT), kernel against oracle from byte-identical arenas."""
import math

import pytest

pytestmark = pytest.mark.gpu
TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"


class Entity:
    """Stand-in for isaaclab.managers.SceneEntityCfg (name + body / joint name patterns -> ids)."""

    def __init__(self, name, body_names=None, joint_names=None):
        self.name, self.body_names, self.joint_names = name, body_names, joint_names
        self.body_ids = self.joint_ids = slice(None)

    def resolve(self, scene):
        ent = scene[self.name]
        if self.body_names is not None:
            self.body_ids = ent.find_bodies(self.body_names)[0]
        if self.joint_names is not None:
            self.joint_ids = ent.find_joints(self.joint_names)[0]


def slipping_feet(env, threshold, asset_cfg, sensor_cfg):
    import torch

    hist = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history  # (N, 3, B, 3)
    touching = hist[:, :, sensor_cfg.body_ids].norm(dim=-1).amax(dim=1) > threshold
    speed = env.scene[asset_cfg.name].data.body_lin_vel_w[:, asset_cfg.body_ids, :2].norm(dim=2)
    return (touching * speed).sum(dim=1)


def dragging_feet(env, asset_cfg, height_threshold, foot_vel_xy_threshold):
    d = env.scene[asset_cfg.name].data
    moving = d.body_lin_vel_w[:, asset_cfg.body_ids, :2].norm(dim=2) > foot_vel_xy_threshold
    low = d.body_pos_w[:, asset_cfg.body_ids, 2] <= height_threshold
    return (moving & low).sum(dim=1)


def leg_links_touching(env, threshold, sensor_cfg):
    hist = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history
    return (hist[:, :, sensor_cfg.body_ids].norm(dim=-1).amax(dim=1) > threshold).sum(dim=1)


def action_change(env):
    term = env.action_manager.get_term("joint_pos")
    return (term.raw_actions - term.prev_raw_actions).square().sum(dim=1)


def object_lifted_off(env, sensor_cfg):
    d = env.scene.sensors[sensor_cfg.name].data
    return ((d.last_contact_time[:, sensor_cfg.body_ids] > 0.0) & (d.current_air_time[:, sensor_cfg.body_ids] > 0.0)).reshape(-1)


def object_in_danger(env, x_max, y_max, z_min, roll_pitch_max, vel_xy_max):
    import torch

    from locotouch_amd.compat import math as M

    r, o = env.scene["robot"].data, env.scene["object"].data
    rel = M.quat_apply_inverse(r.root_quat_w, o.root_pos_w - r.root_pos_w)
    bad = (rel[:, 0].abs() > x_max) | (rel[:, 1].abs() > y_max) | (rel[:, 2] < z_min)
    if roll_pitch_max is not None:
        bad |= torch.acos(-o.projected_gravity_b[:, 2]).abs() > roll_pitch_max * math.pi / 180
    relv = M.quat_apply_inverse(r.root_quat_w, o.root_lin_vel_w - r.root_lin_vel_w)
    return bad | (relv[:, :2].norm(dim=1) > vel_xy_max)


def test_terms_written_against_the_reference_contract_equal_the_fused_columns_on_the_hip_env():
    import torch

    from locotouch_amd import _abi
    from locotouch_amd.compat.runtime import ManagedEnv
    from locotouch_amd.compat.scene_views import link_kinematics
    from locotouch_amd.env import LocoTouchVecEnv

    C = _abi.CONSTS
    n = 512
    cfg = _abi.preset_cfg(TASK, num_envs=n, seed=5)
    cfg.debug_terms = 1
    cfg.reward_weight[C["LT_R_OBJECT_Z_CONTACT"]] = -0.01  # (weight 0 in the registration: never evaluated then)
    for i in range(2):
        cfg.push_robot_interval[i] = cfg.push_obj_interval[i] = 1.0e9  # pushes move the root velocity after the reward stage
    vec = LocoTouchVecEnv(TASK, device="cuda:0", cfg=cfg)
    env = ManagedEnv(TASK, None, vec)
    sensor = "robot_contact_senosr"  # (sic)
    terms = {
        "LT_R_FOOT_SLIP": ("u_slip", slipping_feet, {"threshold": float(cfg.foot_slip_threshold), "asset_cfg": Entity("robot", body_names=".*foot"),
                                                    "sensor_cfg": Entity(sensor, body_names=".*foot")}),
        "LT_R_FOOT_DRAGGING": ("u_drag", dragging_feet, {"asset_cfg": Entity("robot", body_names=".*foot"), "height_threshold": float(cfg.foot_drag_height),
                                                        "foot_vel_xy_threshold": float(cfg.foot_drag_vel)}),
        "LT_R_THIGH_CALF_COLLISION": ("u_links", leg_links_touching, {"threshold": float(cfg.thigh_calf_threshold),
                                                                     "sensor_cfg": Entity(sensor, body_names=[".*thigh", ".*calf"])}),
        "LT_R_ACTION_RATE": ("u_rate", action_change, {}),
        "LT_R_OBJECT_Z_CONTACT": ("u_lift", object_lifted_off, {"sensor_cfg": Entity("object_contact_sensor", body_names="Object")}),
        "LT_R_OBJECT_DANGEROUS_STATE": ("u_danger", object_in_danger, {"x_max": 0.125, "y_max": 0.097, "z_min": 0.095, "roll_pitch_max": None, "vel_xy_max": 2.5}),
    }
    for _, (name, func, params) in terms.items():
        env.add_reward_term(name, func, -1.0e-3, params)
    env.add_reward_term("u_danger_tilt", object_in_danger, -1.0e-3, {"x_max": 0.125, "y_max": 0.097, "z_min": 0.095, "roll_pitch_max": 20.0, "vel_xy_max": 2.5})
    assert env.extra.terms[0][3]["sensor_cfg"].body_ids == [13, 14, 15, 16] and env.extra.terms[2][3]["sensor_cfg"].body_ids == list(range(5, 13))
    g = torch.Generator(device="cuda:0").manual_seed(0)
    nonzero = {k: 0 for k in terms}
    tilted = 0
    for t in range(50):
        act = (0.9 if t % 20 < 12 else 0.05) * torch.randn(n, 12, device="cuda:0", generator=g)
        _, _, dones, _ = env.step(act)
        fused = vec.field("LT_F_REWARD_TERMS").reshape(n, -1)
        keep = dones == 0
        for k, (name, _, _) in terms.items():
            got, want = env.extra.last_values[name][keep], fused[keep, C[k]]
            torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5, msg=lambda m, k=k, t=t: f"{k} at step {t}: {m}")
            nonzero[k] += int((want != 0).sum())
        base, tilt = env.extra.last_values["u_danger"], env.extra.last_values["u_danger_tilt"]
        assert bool((tilt >= base).all())
        tilted += int((tilt != base).sum())
    assert all(v > 0 for v in nonzero.values()), nonzero
    assert tilted > 0, "roll_pitch_max must have fired on its own somewhere"
    # body views: FK of the views against the kernel's own foot kinematics
    f = vec.field
    pos, quat, lin, ang = link_kinematics(f("LT_F_ROOT_POS")[:, 0, :3], f("LT_F_ROOT_QUAT")[:, 0, :4], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3],
                                          f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3], f("LT_F_JOINT_POS").reshape(n, 12), f("LT_F_JOINT_VEL").reshape(n, 12))
    keep = f("LT_F_DONES") == 0
    torch.testing.assert_close(pos[keep][:, 13:17], f("LT_F_FOOT_POS_W").permute(0, 2, 1)[keep], atol=5e-6, rtol=0)
    torch.testing.assert_close(lin[keep][:, 13:17], f("LT_F_FOOT_VEL_W").permute(0, 2, 1)[keep], atol=5e-5, rtol=1e-4)
    d = env.scene["robot"].data
    assert d.body_pos_w.shape == (n, 17, 3) and d.body_quat_w.shape == (n, 17, 4) and d.body_lin_vel_w.shape == (n, 17, 3)
    te = env.extra.env
    assert te.reward_manager.get_term_cfg("gait").func.valid_last_air_time.shape == (n, 4)
    assert te.reward_manager._episode_sums["track_lin_vel_xy"].shape == (n,) and te.termination_manager.terminated.dtype == torch.bool
    assert te.common_step_counter in (50, 51)


def test_user_time_out_requests_match_the_oracle_and_end_the_env_by_time_out():
    import torch

    from locotouch_amd import _abi
    from locotouch_amd.env import LocoTouchVecEnv
    from tests import oracle_lib
    from tests.parity_util import compare_arenas

    C = _abi.CONSTS
    for n in (128, 8208):  # helper form / one-wave form of the step kernel
        env = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=9, debug_terms=1)
        ora = oracle_lib.OracleEnv(env.cfg)
        ora.reset_all()
        g = torch.Generator().manual_seed(3)
        for t in range(3):
            act = 0.5 * torch.randn(n, 12, generator=g)
            env._arena_aligned.copy_(torch.from_numpy(ora.arena))  # identical start state
            want = torch.zeros(n, dtype=torch.bool)
            want[t::5] = True
            both = torch.zeros(n, dtype=torch.bool)
            both[t::35] = True  # some envs carry a termination request as well: `terminated` and `time_out` both set
            env.request_termination(want.to("cuda:0"), time_out=True)
            env.request_termination(both.to("cuda:0"))
            ora.arena[:] = env._arena_aligned.cpu().numpy()  # the request bits travel with the bytes
            env.step(act.to("cuda:0"))
            ora.step(act.numpy())
            torch.cuda.synchronize()
            compare_arenas(env, ora, what=f"n={n} step {t} with time-out requests", max_flip_frac=0.05, max_event_frac=max(2.0 / n, 1e-3))
            bits = env.field("LT_F_TERM_BITS").cpu()
            assert torch.equal(((bits >> C["LT_T_USER_TIME_OUT"]) & 1).bool(), want)
            assert not bool(((bits >> C["LT_TIMEOUT_REQUEST_BIT"]) & 1).any())
            assert bool((env.field("LT_F_TIME_OUT").cpu()[want] != 0).all()) and bool((env.field("LT_F_DONES").cpu()[want] != 0).all())
            only = want & ((bits & ~(1 << C["LT_T_USER_TIME_OUT"]) & 0xFFFF) == 0)
            assert int(only.sum()) > 0 and not bool(env.field("LT_F_TERMINATED").cpu()[only].any())
            assert bool((env.field("LT_F_TERMINATED").cpu()[both] != 0).all()) and bool((env.episode_length_buf.cpu()[want] == 0).all())
