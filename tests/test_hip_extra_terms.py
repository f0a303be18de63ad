"""GPU twin of tests/test_extra_terms.py: the IsaacLab-layout views (compat/scene_views.py) over the HIP env's quad arrays, and a
user reward term added to the kernel's reward through ManagedEnv.add_reward_term (slow path)."""
import pytest

pytestmark = pytest.mark.gpu
TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"


def test_views_and_a_user_reward_term_over_the_hip_env():
    import torch

    from locotouch_amd.compat.runtime import ManagedEnv
    from locotouch_amd.env import LocoTouchVecEnv

    n = 256
    vec = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=4)
    twin = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=4)
    env = ManagedEnv(TASK, None, vec)

    def user_term(env_, threshold: float):  # joint-velocity cost + thigh/calf contacts, read the way the reference's terms read them
        d = env_.scene["robot"].data
        f = env_.scene.sensors["contact_forces"].data.net_forces_w_history
        hit = (torch.max(torch.norm(f[:, :, 5:13], dim=-1), dim=1)[0] > threshold).float().sum(1)
        return torch.sum(torch.square(d.joint_vel), dim=1) + hit + d.root_lin_vel_b[:, 2].abs()

    env.add_reward_term("user", user_term, -1.0e-3, {"threshold": 1.0})
    g = torch.Generator(device="cuda:0").manual_seed(0)
    changed = 0
    for _ in range(25):
        act = 0.8 * torch.randn(n, 12, device="cuda:0", generator=g)
        obs, rew, dones, _ = env.step(act)
        _, rew0, dones0, _ = twin.step(act)
        assert torch.equal(dones, dones0)
        d = env.scene["robot"].data
        jv = vec.field("LT_F_JOINT_VEL")
        for k in range(3):
            for leg in range(4):
                assert torch.equal(d.joint_vel[:, k * 4 + leg], jv[:, k, leg])
        fh = vec.field("LT_F_FORCE_HIST").reshape(n, 3, 4, 4)
        f = env.scene.sensors["contact_forces"].data.net_forces_w_history
        assert torch.equal(torch.norm(f[:, :, 1:17], dim=-1), fh.reshape(n, 3, 16))
        q, v = vec.field("LT_F_ROOT_QUAT")[:, 0, :4], vec.field("LT_F_ROOT_LIN_VEL_W")[:, 0, :3]
        from locotouch_amd.compat import math as M
        assert torch.allclose(d.root_lin_vel_b, M.quat_apply_inverse(q, v))
        keep = (dones == 0).float()
        hit = (fh[:, :, 1:3, :].reshape(n, 3, 8).max(dim=1)[0] > 1.0).float().sum(1)
        want = rew0 + keep * 0.02 * -1.0e-3 * ((jv.reshape(n, 12) ** 2).sum(1) + hit + d.root_lin_vel_b[:, 2].abs())
        torch.testing.assert_close(rew, want, rtol=1e-5, atol=1e-6)
        changed += int((rew != rew0).sum())
    assert changed > n


def test_a_requested_termination_matches_the_oracle_and_resets_in_the_next_step():
    """LT_T_USER (include/lt_env.h): envs whose LT_F_TERM_BITS word carries LT_TERM_REQUEST_BIT terminate in the next step - HIP
    kernel (both forms) and oracle from byte-identical arenas - and a user termination term attached through ManagedEnv drives it."""
    import numpy as np
    import torch

    from locotouch_amd import _abi
    from locotouch_amd.compat.runtime import ManagedEnv
    from locotouch_amd.env import LocoTouchVecEnv
    from tests import oracle_lib
    from tests.parity_util import compare_arenas

    C = _abi.CONSTS
    for n in (128, 8208):  # helper form / one-wave form of the step kernel
        env = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=9, debug_terms=1)
        ora = oracle_lib.OracleEnv(env.cfg)
        ora.reset_all()
        g = torch.Generator().manual_seed(3)
        for t in range(4):
            act = 0.5 * torch.randn(n, 12, generator=g)
            env._arena_aligned.copy_(torch.from_numpy(ora.arena))  # identical start state
            want = torch.zeros(n, dtype=torch.bool)
            want[t::7] = True
            env.request_termination(want.to("cuda:0"))
            ora.arena[:] = env._arena_aligned.cpu().numpy()  # the request bits travel with the bytes
            env.step(act.to("cuda:0"))
            ora.step(act.numpy())
            torch.cuda.synchronize()
            compare_arenas(env, ora, what=f"n={n} step {t} with termination requests", max_flip_frac=0.05, max_event_frac=max(2.0 / n, 1e-3))  # (the allowances of test_hip_parity.py)
            bits = env.field("LT_F_TERM_BITS").cpu()
            assert torch.equal(((bits >> C["LT_T_USER"]) & 1).bool(), want) and not bool(((bits >> C["LT_TERM_REQUEST_BIT"]) & 1).any())
            assert bool((env.field("LT_F_DONES").cpu()[want] != 0).all()) and bool((env.episode_length_buf.cpu()[want] == 0).all())
    # the slow path end to end: a term on the views, one step of delay
    n = 256
    vec = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=4)
    env = ManagedEnv(TASK, None, vec)
    env.add_termination_term("tilted", lambda e, limit: e.scene["robot"].data.projected_gravity_b[:, 2] > -limit, {"limit": 0.985})
    g = torch.Generator(device="cuda:0").manual_seed(0)
    pending = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    total = 0
    for _ in range(30):
        _, _, dones, _ = env.step(0.8 * torch.randn(n, 12, device="cuda:0", generator=g))
        bits = vec.field("LT_F_TERM_BITS")
        user = ((bits >> C["LT_T_USER"]) & 1).bool()
        assert torch.equal(user, pending) and bool((dones[user] != 0).all())
        pending = ((bits >> C["LT_TERM_REQUEST_BIT"]) & 1).bool()
        total += int(user.sum())
    assert total > 0
    _ = np


def test_a_user_observation_term_rides_behind_the_kernels_rows_on_the_gpu():
    """ManagedEnv.add_observation_term over the HIP env: the kernel's rows untouched in front, the user term's history (3 frames,
    oldest -> newest, restarted at a reset) behind them; the fused rollout is bypassed."""
    import torch

    from locotouch_amd.compat.runtime import ManagedEnv, RslRlVecEnvWrapper
    from locotouch_amd.env import LocoTouchVecEnv

    n = 128
    vec = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=2, max_episode_length=8)
    env = ManagedEnv(TASK, None, vec)
    env.add_observation_term("policy", "base_z", lambda e: e.scene["robot"].data.root_pos_w[:, 2:3], history_length=3, scale=10.0)
    w = RslRlVecEnvWrapper(env)
    assert w.num_obs == 348 + 3 and w.num_privileged_obs == 348 and w.fused_target() is None
    obs, _ = env.reset()
    z = vec.field("LT_F_ROOT_POS")[:, 0, 2].clone()
    hist = [z, z, z]
    assert torch.allclose(obs[:, 348:], 10.0 * torch.stack(hist, 1))
    g = torch.Generator(device="cuda:0").manual_seed(0)
    resets = 0
    for _ in range(20):
        obs, rew, dones, ex = env.step(torch.randn(n, 12, device="cuda:0", generator=g))
        z = vec.field("LT_F_ROOT_POS")[:, 0, 2].clone()
        fin = dones != 0
        resets += int(fin.sum())
        hist = [torch.where(fin, z, h) for h in hist[1:] + [z]]
        assert obs.shape == (n, 351) and torch.equal(obs[:, :348], vec.obs_policy)
        assert torch.allclose(obs[:, 348:], 10.0 * torch.stack(hist, 1))
        assert ex["observations"]["critic"].shape == (n, 348)
    assert resets > n
