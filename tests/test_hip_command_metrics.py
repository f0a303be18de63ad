"""C2 on the GPU: `LocoTouchVecEnv.episode_log()` reports the command term's metrics with the reference's per-reset-batch semantics
(locotouch/mdp/commands.py:392-417 + IsaacLab CommandTerm.reset [DEP]) from what the step kernel keeps (LT_F_LAST_CMD_METRICS).  The
kernel's values themselves are compared with the oracle's in every parity step (tests/parity_util.py); the formulas against numpy in
tests/test_oracle_command_metrics.py."""
import pytest

pytestmark = pytest.mark.gpu


def test_episode_log_groups_the_reset_snapshots_by_step():
    import torch

    from locotouch_amd.env import make, reset_batch_means

    env = make("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=256, device="cuda:0", seed=5, max_episode_length=40)
    g = torch.Generator().manual_seed(0)
    env.episode_log()  # opens the window
    resets = torch.zeros(256, dtype=torch.int64)
    for _ in range(30):
        _, _, dones, _ = env.step((1.2 * torch.randn(256, 12, generator=g)).cuda())
        resets += dones.cpu()
    log = env.episode_log()
    torch.cuda.synchronize()
    assert int(resets.sum()) > 10
    last = env.field("LT_F_LAST_CMD_METRICS")[:, 0, :].cpu()
    mask = resets > 0
    want = reset_batch_means(last[mask])
    for name, w in zip(("error_vel_xy", "error_vel_yaw", "foot_air_time_variance"), want):
        assert log[f"Metrics/base_velocity/{name}"] == pytest.approx(w, rel=1e-6, abs=1e-9)
        assert log[f"Metrics/base_velocity/{name}"] >= 0.0
    assert (last[mask, 3] >= 0).all() and (last[~mask, :3] == 0).all()  # envs that never reset hold no snapshot
    cur = env.current_command_metrics()
    assert cur["error_vel_xy"].shape == (256,) and bool(torch.isfinite(cur["error_vel_xy"]).all())
    for key in ("foot_step_frequency", "pair_1_step_frequency", "step_air_time", "lin_vel_x", "rel_standing_envs"):
        assert f"Metrics/base_velocity/{key}" in log
    # a window without resets reports no per-batch means (the reference logs nothing when nobody resets)
    env2 = make("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=64, device="cuda:0", seed=6)
    env2.episode_log()
    env2.step(torch.zeros(64, 12, device="cuda:0"))
    assert "Metrics/base_velocity/error_vel_xy" not in env2.episode_log()


def test_binary_maximal_command_matches_the_oracle_step_by_step():
    """cfg.cmd_binary_maximal (commands.py:518-521) through resets and timer resamples: kernel == oracle from byte-identical arenas,
    and every command is one of the 8 corner commands."""
    import numpy as np
    import torch

    from locotouch_amd.layout import Layout
    from tests import oracle_lib as O
    from tests.parity_util import Tally, compare_arenas
    from tests.test_hip_parity import make_env

    n = 64
    env = make_env("teacher", n, cmd_binary_maximal=1, max_episode_length=12)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    g = torch.Generator().manual_seed(4)
    tally = Tally(n)
    L = Layout(n, env.num_obs, 0)
    resets = 0
    for t in range(40):
        act = torch.randn(n, 12, generator=g)
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        env.step(act.cuda())
        ora.step(act.numpy())
        torch.cuda.synchronize()
        tally.add(compare_arenas(env, ora, what=f"binary maximal step {t}", max_flip_frac=0.05, max_event_frac=2.0 / n))
        resets += int(L.arr(ora.arena, "LT_F_DONES")[:n].sum())
    assert resets > n
    P = env.cmd_params.cpu().numpy()
    buf = env.field("LT_F_CMD_BUF")[:, 0, :3].cpu().numpy()
    assert (np.abs(buf) == np.array([P[1], P[3], P[5]], np.float32)).all()
