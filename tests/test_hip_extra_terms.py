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
