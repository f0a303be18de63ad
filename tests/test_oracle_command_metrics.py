"""C2: the command term's per-env metrics and their per-reset-batch log (locotouch/mdp/commands.py:392-417 `_update_metrics`;
IsaacLab CommandTerm.reset [DEP] logs the mean over the envs that reset in a step of the values the last compute() left and zeroes
them; the runner averages the per-step entries).  CPU: the oracle's restatement against the formulas written out in numpy on the
oracle's own state, with pushes and timer resamples switched off so that the state and command `_update_metrics` saw are the ones
the arena holds after the step.  The HIP kernel is compared with the oracle field by field in every parity step
(tests/parity_util.py: LT_F_EVENT_TIMERS lanes 2-3, LT_F_TRUNK_FORCE_HIST lane 3, LT_F_LAST_CMD_METRICS)."""
import numpy as np
import torch

from locotouch_amd import _abi
from locotouch_amd.env import reset_batch_means
from tests.oracle_vec_env import OracleVecEnv


def _quat_apply_inv(q, v):
    w, u = q[:, :1], q[:, 1:]
    t = 2.0 * np.cross(u, v)
    return v - w * t + np.cross(u, t)


def test_metrics_follow_update_metrics_and_the_reset_batch_log():
    cfg = _abi.preset_cfg("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=48, seed=11)
    cfg.push_robot_interval[0] = cfg.push_robot_interval[1] = 1e6   # no interval pushes: the arena's velocities are the ones compute() saw
    cfg.push_obj_interval[0] = cfg.push_obj_interval[1] = 1e6
    cfg.cmd_resample_time[0] = cfg.cmd_resample_time[1] = 1e6       # no timer resample: a non-reset env keeps its command
    cfg.max_episode_length = 23
    env = OracleVecEnv("", cfg=cfg)
    n = env.num_envs
    f = lambda name: env.field(name).numpy()  # noqa: E731
    assert (f("LT_F_EVENT_TIMERS")[:, 0, 2:] == 0).all() and (f("LT_F_LAST_CMD_METRICS") == 0).all()  # reset() runs no compute()
    g = torch.Generator().manual_seed(2)
    prev_cmd = f("LT_F_CMD")[:, 0, :3].copy()
    prev_m = np.zeros((n, 3), np.float32)
    per_step_logs, last_reset_step, history = [], np.full(n, -1), []
    for t in range(70):
        step_id = int(env.field("LT_F_COUNTERS")[0])
        _, _, dones, _ = env.step(1.5 * torch.randn(n, 12, generator=g))
        done = dones.numpy().astype(bool)
        q, v, w = f("LT_F_ROOT_QUAT")[:, 0, :], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3], f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3]
        # the command _update_metrics saw: a reset env's fresh sample (kept in the buffer; the zero-command window only blanks
        # vel_command_b afterwards), anybody else's command of the step before
        cmd = np.where(done[:, None], f("LT_F_CMD_BUF")[:, 0, :3], prev_cmd)
        vb, wb = _quat_apply_inv(q, v), _quat_apply_inv(q, w)
        air = f("LT_F_FOOT_LAST_AIR")[:, 0, :]
        want = np.stack([np.linalg.norm(cmd[:, :2] - vb[:, :2], axis=1), np.abs(cmd[:, 2] - wb[:, 2]), air.var(axis=1, ddof=1)], axis=1)
        got = np.concatenate([f("LT_F_EVENT_TIMERS")[:, 0, 2:], f("LT_F_TRUNK_FORCE_HIST")[:, 0, 3:]], axis=1)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        if done.any():  # CommandTerm.reset: the batch mean of what the LAST compute() left, i.e. the values of the step before
            last = f("LT_F_LAST_CMD_METRICS")[:, 0, :]
            np.testing.assert_array_equal(last[done, :3], prev_m[done])
            assert (last[done, 3] == step_id).all()
            per_step_logs.append(prev_m[done].mean(axis=0))
            last_reset_step[done] = step_id
        history.append((done.copy(), prev_m.copy()))
        prev_cmd, prev_m = f("LT_F_CMD")[:, 0, :3].copy(), got.astype(np.float32)
    assert len(per_step_logs) >= 3 and (last_reset_step >= 0).sum() > n // 2
    # A log window as the runner forms it (mean over the window's per-step batch means) == the host-side grouping of the snapshots
    # (reset_batch_means), over the envs that reset exactly once in the window (a later reset replaces the earlier snapshot)
    window = history[-18:]
    resets = np.sum([d for d, _ in window], axis=0)
    once = resets == 1
    assert once.sum() >= 8
    want = np.mean([m[d & once].mean(axis=0) for d, m in window if (d & once).any()], axis=0)
    rows = torch.from_numpy(f("LT_F_LAST_CMD_METRICS")[:, 0, :][once].copy())
    np.testing.assert_allclose(reset_batch_means(rows), want, rtol=1e-5)


def test_binary_maximal_command_draws_the_eight_corner_commands():
    """`binary_maximal_command` (commands.py:95-104, 189-197, 518-521; off in every registered config): a resample picks one of the 8
    sign combinations uniformly and scales it by the current upper range bounds; `is_standing_env` is left alone."""
    cfg = _abi.preset_cfg("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=512, seed=3)
    cfg.cmd_binary_maximal = 1
    cfg.cmd_zero_steps = 0
    env = OracleVecEnv("", cfg=cfg)
    P = env.cmd_params.numpy()
    hi = np.array([P[1], P[3], P[5]], np.float32)
    cmd = env.field("LT_F_CMD_BUF").numpy()[:, 0, :3]
    assert (np.abs(cmd) == hi).all()
    combos = {tuple(np.sign(c).astype(int)) for c in cmd}
    assert len(combos) == 8
    counts = np.array([sum(1 for c in cmd if tuple(np.sign(c).astype(int)) == k) for k in sorted(combos)])
    assert counts.min() > 512 / 8 * 0.5 and counts.max() < 512 / 8 * 1.6  # uniform over the 8 corners
    assert (env.field("LT_F_CMD_BUF").numpy()[:, 0, 3] == 0).all()  # nobody was made a standing env by the resample
