"""SURVEY.md §8(b) B3, slow path: a reward term the fused kernels do not know no longer stops a cfg - it is evaluated in torch on
IsaacLab-layout views of the env state (compat/scene_views.py) and added to the kernel's reward.  CPU: the reference's own env cfg
with one extra user term, translated, on the oracle env, trained for an iteration.  (GPU twin: tests/test_hip_extra_terms.py.)"""
import os
import sys

import numpy as np
import pytest
import torch

from locotouch_amd import _abi

REF = "/root/reference"
TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def joint_vel_l2_user(env, asset_cfg):
    """a user term in the reference's style (mdp/rewards.py:442-444): reads asset.data through a SceneEntityCfg"""
    asset = env.scene[asset_cfg.name]
    return torch.sum(torch.square(asset.data.joint_vel[:, asset_cfg.joint_ids]), dim=1)


def feet_force_user(env, sensor_cfg, threshold: float):
    """reads the contact sensor's force history (mdp/rewards.py:459-466 style)"""
    forces = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history
    return torch.sum((torch.max(torch.norm(forces[:, :, sensor_cfg.body_ids], dim=-1), dim=1)[0] > threshold).float(), dim=1)


@pytest.fixture(scope="module")
def rt():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings

    from locotouch_amd.compat import runtime

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        runtime.install()
        import locotouch  # noqa: F401
    return runtime


def test_reference_cfg_with_a_user_term_translates_steps_and_trains(rt, tmp_path):
    from isaaclab.managers import RewardTermCfg as RewTerm
    from isaaclab.managers import SceneEntityCfg

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.compat import cfg_translate as T
    from locotouch_amd.compat.runtime import ManagedEnv, RslRlVecEnvWrapper
    from locotouch_amd.rl import OnPolicyRunner
    from tests.oracle_vec_env import OracleVecEnv

    cfg = rt.load_cfg_from_registry(TASK, "env_cfg_entry_point")
    cfg.scene.num_envs = 48
    cfg.rewards.user_joint_vel = RewTerm(func=joint_vel_l2_user, weight=-2.0e-3, params={"asset_cfg": SceneEntityCfg("robot", joint_names=".*_calf_joint")})
    cfg.rewards.user_feet = RewTerm(func=feet_force_user, weight=0.25, params={"sensor_cfg": SceneEntityCfg("robot_contact_senosr", body_names=".*foot"), "threshold": 1.0})
    with pytest.raises(T.UnsupportedCfg):
        T.translate(cfg)  # the strict form still refuses
    lt, sizes = rt.translate_env_cfg(TASK, cfg)
    assert [t[0] for t in lt.extra_reward_terms] == ["user_joint_vel", "user_feet"]
    vec = OracleVecEnv(TASK, cfg=lt, object_sizes=sizes)
    plain = OracleVecEnv(TASK, cfg=lt, object_sizes=sizes)  # the same env without the slow path
    env = ManagedEnv(TASK, cfg, vec, extra_rewards=lt.extra_reward_terms)
    assert env.extra.terms[0][3]["asset_cfg"].joint_ids == [8, 9, 10, 11] and env.extra.terms[1][3]["sensor_cfg"].body_ids == [13, 14, 15, 16]
    g = torch.Generator().manual_seed(0)
    seen = 0
    for _ in range(30):
        act = 0.8 * torch.randn(48, 12, generator=g)
        obs, rew, dones, _ = env.step(act)
        _, rew0, dones0, _ = plain.step(act)
        assert torch.equal(dones, dones0)
        keep = (dones == 0).float()
        jv = vec.field("LT_F_JOINT_VEL")[:, 2, :]  # calf joints of the four legs
        fh = vec.field("LT_F_FORCE_HIST").reshape(48, 3, 4, 4)[:, :, 3, :]  # foot |F| history
        want = rew0 + keep * env.extra.env.step_dt * (-2.0e-3 * (jv ** 2).sum(1) + 0.25 * (fh.max(dim=1)[0] > 1.0).float().sum(1))
        torch.testing.assert_close(rew, want, rtol=1e-5, atol=1e-6)
        seen += int((rew != rew0).sum())
    assert seen > 48, "the user terms must have contributed"
    # and the trainer runs on it
    w = RslRlVecEnvWrapper(env)
    agent = train_cfg(TASK)
    agent["num_steps_per_env"] = 4
    runner = OnPolicyRunner(w, agent, log_dir=None, device="cpu")
    runner.learn(num_learning_iterations=1)
    assert all(torch.isfinite(p).all() for p in runner.alg.actor_critic.parameters())


def test_views_follow_the_isaaclab_layouts():
    from locotouch_amd.compat.scene_views import BODY_NAMES, JOINT_NAMES, TermEnv
    from tests.oracle_vec_env import OracleVecEnv

    vec = OracleVecEnv(TASK, num_envs=16, seed=1)
    for _ in range(5):
        vec.step(torch.zeros(16, 12))
    te = TermEnv(vec)
    d = te.scene["robot"].data
    assert d.joint_pos.shape == (16, 12) and d.soft_joint_pos_limits.shape == (16, 12, 2) and d.root_state_w.shape == (16, 13)
    jp = vec.field("LT_F_JOINT_POS")
    for k in range(3):
        for leg in range(4):
            assert torch.equal(d.joint_pos[:, k * 4 + leg], jp[:, k, leg])
    assert JOINT_NAMES[4] == "a_FR_thigh_joint" and BODY_NAMES[13:] == ["a_FR_foot", "b_FL_foot", "c_RR_foot", "d_RL_foot"]
    np.testing.assert_allclose(d.projected_gravity_b.norm(dim=1).numpy(), 1.0, atol=1e-5)
    assert torch.allclose(d.default_joint_pos[0, :4], torch.tensor([-0.1, 0.1, -0.1, 0.1]))
    c = te.scene.sensors["robot_contact_senosr"].data
    assert c.net_forces_w_history.shape == (16, 3, 17, 3) and c.current_air_time.shape == (16, 17)
    fh = vec.field("LT_F_FORCE_HIST").reshape(16, 3, 4, 4)
    assert torch.equal(torch.norm(c.net_forces_w_history[:, :, 13:17], dim=-1), fh[:, :, 3, :])
    assert torch.equal(c.force_norm_history[:, :, 0], vec.field("LT_F_TRUNK_FORCE_HIST")[:, 0, :3])
    assert te.scene["object"].data.root_pos_w.shape == (16, 3) and te.command_manager.get_command("base_velocity").shape == (16, 3)
    ids, names = te.scene["contact_forces"].find_bodies(".*hip")
    assert ids == [1, 2, 3, 4]


def base_too_fast_user(env, limit: float, asset_cfg):
    """a user termination in the reference's style (mdp/terminations.py:10-23): reads asset.data, returns bool (N,)"""
    asset = env.scene[asset_cfg.name]
    return torch.norm(asset.data.root_lin_vel_w[:, :2], dim=1) > limit


def test_reference_cfg_with_a_user_termination_term_terminates_one_step_later(rt):
    """B3, terminations: the reference's env cfg with one extra TerminationTermCfg translates; the term is evaluated on the state a
    step left and the NEXT step terminates the env through the kernel's own termination stage (LT_T_USER, include/lt_env.h)."""
    from isaaclab.managers import SceneEntityCfg
    from isaaclab.managers import TerminationTermCfg as DoneTerm

    from locotouch_amd.compat import cfg_translate as T
    from locotouch_amd.compat.runtime import ManagedEnv
    from tests.oracle_vec_env import OracleVecEnv

    C = _abi.CONSTS
    cfg = rt.load_cfg_from_registry(TASK, "env_cfg_entry_point")
    cfg.scene.num_envs = 64
    cfg.terminations.base_too_fast = DoneTerm(func=base_too_fast_user, params={"limit": 0.35, "asset_cfg": SceneEntityCfg("robot")})
    with pytest.raises(T.UnsupportedCfg):
        T.translate(cfg)  # the strict form still refuses
    lt, sizes = rt.translate_env_cfg(TASK, cfg)  # (user TIME-OUT terms: tests/test_reference_terms_slow_path.py)
    assert [t[0] for t in lt.extra_termination_terms] == ["base_too_fast"] and not lt.extra_reward_terms
    vec = OracleVecEnv(TASK, cfg=lt, object_sizes=sizes)
    env = ManagedEnv(TASK, cfg, vec, extra_terminations=lt.extra_termination_terms)
    g = torch.Generator().manual_seed(1)
    pending = torch.zeros(64, dtype=torch.bool)
    fired_total = 0
    for _ in range(40):
        act = 0.8 * torch.randn(64, 12, generator=g)
        _, _, dones, _ = env.step(act)
        bits = torch.from_numpy(vec._arr("LT_F_TERM_BITS").copy())
        user = ((bits >> C["LT_T_USER"]) & 1).bool()
        assert torch.equal(user, pending), "exactly the envs requested after the previous step terminate by LT_T_USER in this one"
        assert bool((dones[user] != 0).all()) and bool((torch.from_numpy(vec._arr("LT_F_TERMINATED").copy())[user] != 0).all())
        pending = ((bits >> C["LT_TERM_REQUEST_BIT"]) & 1).bool()  # what the slow path requested on the state this step left
        v = vec.field("LT_F_ROOT_LIN_VEL_W")[:, 0, :2]
        assert torch.equal(pending, (v.norm(dim=1) > 0.35) & (dones == 0))
        fired_total += int(user.sum())
    assert fired_total > 5 and env.extra.term_counts["base_too_fast"] >= fired_total


def base_height_user(env, asset_cfg):
    """a user observation term in IsaacLab's style (isaaclab.envs.mdp.observations.base_pos_z [DEP]): (N, 1)"""
    return env.scene[asset_cfg.name].data.root_pos_w[:, 2].unsqueeze(-1)


def calf_vel_user(env, asset_cfg):
    return env.scene[asset_cfg.name].data.joint_vel[:, asset_cfg.joint_ids]


def test_user_observation_terms_follow_the_observation_manager(rt):
    """VERDICT r03 missing #4: observation terms outside the fused set.  A derived cfg appends two ObsTerms to the policy group and one to
    the critic group; the translated env returns the kernel's rows followed by the user terms' columns with IsaacLab's
    ObservationManager semantics [DEP]: clip, scale, the group's history (6 frames, oldest -> newest, a reset env's history filled
    with its first value), additive uniform noise in the corrupting group only."""
    from isaaclab.managers import ObservationTermCfg as ObsTerm
    from isaaclab.managers import SceneEntityCfg
    from isaaclab.utils.noise import AdditiveUniformNoiseCfg as Unoise

    from locotouch_amd.compat import cfg_translate as T
    from locotouch_amd.compat.runtime import ManagedEnv, RslRlVecEnvWrapper
    from tests.oracle_vec_env import OracleVecEnv

    cfg = rt.load_cfg_from_registry(TASK, "env_cfg_entry_point")
    cfg.scene.num_envs = 24
    cfg.observations.policy.base_height = ObsTerm(func=base_height_user, params={"asset_cfg": SceneEntityCfg("robot")}, scale=2.0, clip=(0.0, 0.3))
    cfg.observations.policy.calf_vel = ObsTerm(func=calf_vel_user, params={"asset_cfg": SceneEntityCfg("robot", joint_names=".*_calf_joint")},
                                               noise=Unoise(n_min=-0.5, n_max=0.5))
    cfg.observations.critic.base_height = ObsTerm(func=base_height_user, params={"asset_cfg": SceneEntityCfg("robot")})
    with pytest.raises(T.UnsupportedCfg):
        T.translate(cfg)
    lt, sizes = rt.translate_env_cfg(TASK, cfg)
    assert [(o["group"], o["name"], o["history_length"]) for o in lt.extra_observation_terms] == [("policy", "base_height", 6), ("policy", "calf_vel", 6), ("critic", "base_height", 6)]
    assert lt.extra_observation_terms[1]["noise"] == (-0.5, 0.5) and lt.extra_observation_terms[2]["noise"] is None
    vec = OracleVecEnv(TASK, cfg=lt, object_sizes=sizes)
    env = ManagedEnv(TASK, cfg, vec, extra_observations=lt.extra_observation_terms)
    w = RslRlVecEnvWrapper(env)
    assert w.num_obs == 348 + 6 * (1 + 4) and w.num_privileged_obs == 348 + 6 and w.fused_target() is None
    obs0, ex0 = env.reset()
    assert obs0.shape == (24, 378) and ex0["observations"]["critic"].shape == (24, 354)
    z0 = vec.field("LT_F_ROOT_POS")[:, 0, 2]
    torch.testing.assert_close(obs0[:, 348:354], (2.0 * z0.clamp(0.0, 0.3)).unsqueeze(1).expand(24, 6))  # the first value fills the history
    hist_z = [z0.clone()] * 6
    g = torch.Generator().manual_seed(1)
    vec.cfg.max_episode_length = vec.o.cfg.max_episode_length = 9  # resets inside the window
    for t in range(14):
        obs, rew, dones, ex = env.step(0.7 * torch.randn(24, 12, generator=g))
        z = vec.field("LT_F_ROOT_POS")[:, 0, 2].clone()
        fin = dones != 0
        hist_z = hist_z[1:] + [z]
        if bool(fin.any()):
            hist_z = [torch.where(fin, z, h) for h in hist_z]
        want = torch.stack(hist_z, dim=1)
        assert torch.equal(obs[:, :348], torch.from_numpy(vec._arr("LT_F_OBS_POLICY")))
        torch.testing.assert_close(obs[:, 348:354], 2.0 * want.clamp(0.0, 0.3))
        torch.testing.assert_close(ex["observations"]["critic"][:, 348:354], want)          # no clip / scale, no noise in the critic group
        newest_calf = obs[:, 354:378].reshape(24, 6, 4)[:, -1]
        clean = vec.field("LT_F_JOINT_VEL")[:, 2, :]
        assert float((newest_calf - clean).abs().max()) <= 0.5 + 1e-6 and float((newest_calf - clean).abs().max()) > 0.05  # noisy, bounded
        o2, e2 = env.get_observations()
        assert torch.equal(o2, obs) and torch.equal(e2["observations"]["critic"], ex["observations"]["critic"])  # no frame pushed by a read
    # and the trainer builds its networks for the wider rows and runs on them
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.rl import OnPolicyRunner

    agent = train_cfg(TASK)
    agent["num_steps_per_env"] = 4
    runner = OnPolicyRunner(w, agent, log_dir=None, device="cpu")
    runner.learn(num_learning_iterations=1)
    assert runner.alg.actor_critic.actor[0].in_features == 378 and runner.alg.actor_critic.critic[0].in_features == 354
    assert all(torch.isfinite(p).all() for p in runner.alg.actor_critic.parameters())
