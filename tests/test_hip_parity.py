"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
against the reference's golden vectors, and - at BASELINE.json's full size - through size-independent properties."""
import os

import numpy as np
import pytest

from locotouch_amd import _abi
from locotouch_amd.layout import Layout
from tests import oracle_lib as O
from tests.parity_util import Tally, compare_arenas, compare_host_arenas, device_arena_to_host

pytestmark = pytest.mark.gpu
C = _abi.CONSTS
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TASKS = {"teacher": "Isaac-RandCylinderTransportTeacher-LocoTouch-v1", "locomotion": "Isaac-Locomotion-LocoTouch-v1",
         "student": "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"}


def make_env(task, n, seed=11, **kw):
    import torch
    from locotouch_amd.env import LocoTouchVecEnv

    env = LocoTouchVecEnv(TASKS[task], num_envs=n, device="cuda:0", seed=seed, debug_terms=1, **kw)
    torch.cuda.synchronize()
    return env


@pytest.mark.parametrize("task,n", [("teacher", 64), ("teacher", 37), ("locomotion", 48), ("student", 53)])
def test_reset_all_matches_oracle(task, n):
    """Startup + reset events (counter-based Philox: same draws on both sides) and the first observation."""
    env = make_env(task, n)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    compare_arenas(env, ora, what=f"reset_all {task}")
    L = Layout(n, env.num_obs, int(env.cfg.tactile_enabled))
    a = device_arena_to_host(env)
    assert (L.arr(a, "LT_F_EP_LEN")[:n] == 0).all() and L.arr(a, "LT_F_COUNTERS")[0] == 1
    obs = L.arr(a, "LT_F_OBS_POLICY")[:n]
    frame = env.num_obs // 6
    assert frame in (45, 58)
    # first push after reset fills all 6 history slots: every term block repeats its newest frame
    dims = [3, 3, 3, 12, 12, 12] + ([13] if task != "locomotion" else [])
    off = 0
    for d in dims:
        blk = obs[:, off:off + 6 * d].reshape(n, 6, d)
        assert (blk == blk[:, -1:, :]).all()
        off += 6 * d


def test_registered_presets_and_explicit_cylinders_match_oracle():
    """The other registrations (fixed cylinder + robot-frame object reset, -Play- sizes, velocity-curriculum locomotion) and
    the explicit per-env cylinder table a translated cfg tree carries: reset + steps against the oracle from identical bytes."""
    import torch
    from locotouch_amd.env import LocoTouchVecEnv, task_ids

    assert len(task_ids()) >= 8
    g = torch.Generator().manual_seed(9)
    cases = [("Isaac-CylinderTransportTeacher-LocoTouch-v1", 48, None), ("Isaac-LocomotionVelCur-LocoTouch-Play-v1", None, None),
             ("Isaac-RandCylinderTransportTeacher-LocoTouch-Play-v1", None, "sizes")]
    for task, n, sizes in cases:
        kw = {}
        if sizes:
            kw["object_sizes"] = torch.stack([0.03 + 0.04 * torch.rand(50, generator=g), 0.1 + 0.3 * torch.rand(50, generator=g)], 1)
        env = LocoTouchVecEnv(task, num_envs=n, device="cuda:0", seed=5, debug_terms=1, **kw)
        n = env.num_envs
        assert n == (48 if "CylinderTransportTeacher-LocoTouch-v1" in task and "Rand" not in task else 50)
        ora = O.OracleEnv(env.cfg)
        L = Layout(n, env.num_obs)
        if sizes:
            L.arr(ora.arena, "LT_F_OBJ_SIZES")[:n] = kw["object_sizes"].numpy()
        ora.reset_all()
        torch.cuda.synchronize()
        compare_arenas(env, ora, what=f"reset_all {task}")
        if sizes:
            np.testing.assert_array_equal(L.vec(ora.arena, "LT_F_OBJ_PARAMS")[:, :2], kw["object_sizes"].numpy())
        if "CylinderTransportTeacher-LocoTouch-v1" in task and "Rand" not in task:
            assert env.cfg.obj_reset_robot_frame == 1 and (L.vec(ora.arena, "LT_F_OBJ_PARAMS")[:, 0] == np.float32(0.05)).all()
        tally = Tally(n)
        for t in range(30):
            act = (0.4 if t > 5 else 0.0) * torch.randn(n, 12, generator=g)
            env._arena_aligned.copy_(torch.from_numpy(ora.arena))
            env.step(act.cuda())
            ora.step(act.numpy())
            torch.cuda.synchronize()
            tally.add(compare_arenas(env, ora, what=f"{task} step {t}", max_flip_frac=0.05, max_event_frac=2.0 / n))
        print(tally.line(task))
        assert tally.flips <= 0.01 * n * 30 and tally.events <= 0.005 * n * 30


@pytest.mark.parametrize("task,n,steps,phys,pre", [
    ("teacher", 64, 160, 1, 0), ("locomotion", 64, 120, 1, 0), ("teacher", 32, 40, 2, 0),
    # student task: + tactile refresh cadence, plate samples, lt_tactile_kernel rows (405 = the registration's env count)
    ("student", 405, 100, 1, 0),
    # > 8192 envs: launch_step picks the register-path (PREFETCH=false) history variant, which shifts the rows in place
    ("teacher", 8208, 16, 1, 45), ("locomotion", 8208, 12, 1, 45),
    # the headline grid itself: 4096 envs = 256 tiles, one per CU, four-wave form across all XCDs
    ("teacher", 4096, 16, 1, 45), ("locomotion", 4096, 12, 1, 45)])
def test_step_parity_resynced(task, n, steps, phys, pre):
    """Every step starts from byte-identical state (oracle arena copied to the device), then one step on each side.
    Covers contacts, object resting/rolling, resets with RNG, command resampling, pushes, history shifting.
    `pre`: oracle-only warm-up steps (OpenMP) so that a short compared window still holds resets and shifted histories."""
    import torch

    env = make_env(task, n, phys_substeps=phys)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    g = torch.Generator().manual_seed(3)
    for _ in range(pre):
        ora.step((0.6 * torch.randn(n, 12, generator=g)).numpy(), nthreads=8)
    n_reset, n_shift, lit = 0, 0, 0
    tally = Tally(n)
    L = Layout(n, env.num_obs, int(env.cfg.tactile_enabled))
    for t in range(steps):
        scale = 1.0 if pre else (0.0 if t < 10 else (0.3 if t < steps // 2 else 1.0))
        act = scale * torch.randn(n, 12, generator=g)
        if t % 17 == 0:
            act[0, 0] = 400.0  # exercises the +-100 raw clip
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        env.step(act.cuda())
        ora.step(act.numpy(), nthreads=8 if n > 1024 else 1)
        torch.cuda.synchronize()
        res = compare_arenas(env, ora, what=f"{task} n={n} step {t}", max_flip_frac=0.05, max_event_frac=max(2.0 / n, 1e-3))
        tally.add(res)
        d = L.arr(ora.arena, "LT_F_DONES")[:n]
        n_reset += int(d.sum())
        n_shift += int((d == 0).sum())
        if task == "student":
            ex = env.get_observations()[1]["observations"]
            assert set(ex) == {"policy", "critic", "tactile", "object_state"}
            assert ex["tactile"].shape == (n, 442) and ex["object_state"].shape == (n, 78)
            assert torch.equal(ex["object_state"], ex["policy"][:, 270:])  # a zero-copy window of the policy rows
            lit += int(ex["tactile"][:, :221].sum())
    print(tally.line(f"{task} n={n} phys={phys}"))
    if task == "student":
        assert lit > 3 * n * steps // 2, "the carried cylinders must light taxels"
    assert n_reset > 0, "the sequence must include resets"
    assert n_shift > n_reset, "most rows must really shift (not fill)"
    assert tally.flips <= 0.01 * n * steps, f"too many thresholded-contact flips: {tally.flips}"
    # observed: 1.5e-4 of the env-steps (r02: 14-19 events in 98-131 k env-steps at 8208 envs); the bound sits near it
    assert tally.events <= max(2, 5e-4 * n * steps), f"too many discontinuous-event divergences: {tally.events}"


@pytest.mark.parametrize("fmt,aux", [("LT_TACTILE_PROCESSED", 3), ("LT_TACTILE_DISCRETE", 0), ("LT_TACTILE_NORMALIZED", 1),
                                     ("LT_TACTILE_CONTINUOUS", 2), ("LT_TACTILE_ORIGINAL", 0)])
def test_tactile_formats_and_play_groups_match_oracle(fmt, aux):
    """O6 + the student -Play- env's groups: every TactileSignals class through lt_tactile_kernel against the oracle (itself pinned
    to the reference's classes by replayed uniforms, tests/test_oracle_replay.py), resynced every step: contact channels exact up to
    threshold flips, force channels in the fp32 band (parity_util)."""
    import torch

    from locotouch_amd import _abi

    C = _abi.CONSTS
    n, steps = 64, 45
    env = make_env("student", n, tactile_format=C[fmt], tactile_aux_groups=aux)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    wide = fmt in ("LT_TACTILE_PROCESSED", "LT_TACTILE_ORIGINAL")
    g = torch.Generator().manual_seed(5)
    tally = Tally(n)
    seen = {k: 0.0 for k in ("contact", "second", "levels")}
    for t in range(steps):
        act = (0.0 if t < 10 else 0.5) * torch.randn(n, 12, generator=g)
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        env.step(act.cuda())
        ora.step(act.numpy())
        torch.cuda.synchronize()
        tally.add(compare_arenas(env, ora, what=f"{fmt} aux {aux} step {t}", max_flip_frac=0.05, max_event_frac=2.0 / n))
        ex = env.get_observations()[1]["observations"]
        want = {"policy", "critic", "tactile", "object_state"} | ({"original_tactile"} if aux & 1 else set()) | ({"processed_tactile"} if aux & 2 else set())
        assert set(ex) == want and ex["tactile"].shape == (n, 884 if wide else 442)
        for k in ("original_tactile", "processed_tactile"):
            if k in ex:
                assert ex[k].shape == (n, 884) and ex[k].is_contiguous()
        tac = ex["tactile"]
        seen["contact"] += float(tac[:, :221].sum())
        seen["second"] += float(tac[:, 221:442].sum())
        lv = tac[:, -221:][tac[:, :221] > 0]
        seen["levels"] = max(seen["levels"], float(lv.max()) if lv.numel() else 0.0)
    print(tally.line(f"{fmt} aux {aux}"))
    assert seen["contact"] > n * steps and seen["second"] > 0 and 0 < seen["levels"] <= 1.0
    assert tally.flips <= 0.02 * n * steps and tally.events <= 0.005 * n * steps


def test_global_gate_kernel_matches_oracle():
    """Multi-rank curriculum gate (cfg.cur_gate_external): the step kernel publishes its population sums into LT_F_GATE_RING and
    leaves the widening to lt_env_curriculum_apply_global; HIP and oracle, fed the same (here: single-rank) sums, must keep
    identical command blocks, and the ranges must really widen."""
    import torch
    from locotouch_amd.rl import Dist

    n, rollout = 64, 8
    from locotouch_amd.env import LocoTouchVecEnv

    cfg = _abi.preset_cfg(TASKS["teacher"], num_envs=n, seed=11)
    cfg.cur_gate_external, cfg.env_index_offset, cfg.debug_terms = 1, 4096, 1
    cfg.max_episode_length, cfg.cur_len_threshold = 12, 2.0  # every env times out often: whole populations pass the gate quickly
    cfg.cur_reward_threshold[0] = cfg.cur_reward_threshold[1] = -1.0e3
    env = LocoTouchVecEnv(TASKS["teacher"], device="cuda:0", cfg=cfg)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    torch.cuda.synchronize()
    compare_arenas(env, ora, what="reset_all (offset RNG keys)")
    L = Layout(n, env.num_obs)
    g = torch.Generator().manual_seed(5)
    dist = Dist()
    bins = []
    for it in range(30):
        for t in range(rollout):
            act = 0.3 * torch.randn(n, 12, generator=g)
            env._arena_aligned.copy_(torch.from_numpy(ora.arena))
            env.step(act.cuda())
            ora.step(act.numpy())
            torch.cuda.synchronize()
            compare_arenas(env, ora, what=f"external gate it {it} step {t}", max_flip_frac=0.05, max_event_frac=2.0 / n)
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        env.curriculum_sync(dist, rollout)
        ring = L.arr(ora.arena, "LT_F_GATE_RING").copy()
        ora.curriculum_apply_global(ring, rollout, n)
        torch.cuda.synchronize()
        Pd = device_arena_to_host(env)
        np.testing.assert_array_equal(L.arr(Pd, "LT_F_CMD_PARAMS"), L.arr(ora.arena, "LT_F_CMD_PARAMS"))
        bins.append(float(L.arr(ora.arena, "LT_F_CMD_PARAMS")[17] + L.arr(ora.arena, "LT_F_CMD_PARAMS")[18]))
    assert bins[-1] >= 3, bins
    assert int(L.arr(ora.arena, "LT_F_COUNTERS")[3]) == 30 * rollout


def test_chained_steps_at_the_headline_grid_match_the_oracle():
    """4096 envs (one tile per CU), lt_env_defer_gate mode 2: chains of 1-3 steps from byte-identical state against the ORACLE
    (which runs its population pass behind every step) - compared after each chain has been closed by gate_update().  A chain is
    not resynced inside, so a k-step chain is compared at k times the one-step event allowance."""
    import torch

    n = 4096
    env = make_env("teacher", n)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    g = torch.Generator().manual_seed(11)
    for _ in range(30):
        ora.step((0.6 * torch.randn(n, 12, generator=g)).numpy(), nthreads=8)
    tally = Tally(n)
    L = Layout(n, env.num_obs)
    off = L.plain["LT_F_COUNTERS"][0] + 16  # counters[2]: the chain's scratch flag
    nsteps = 0
    for chain in (1, 2, 1, 3, 2, 1):
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        env.defer_gate(2)
        for _ in range(chain):
            act = 0.6 * torch.randn(n, 12, generator=g)
            a_dev = act.cuda()
            env.step_rows_raw(a_dev.data_ptr(), 0, 0, 0, 0)
            ora.step(act.numpy(), nthreads=8)
            nsteps += 1
        env.gate_update()
        env.defer_gate(0)
        torch.cuda.synchronize()
        dev = device_arena_to_host(env)
        dev[off:off + 8] = ora.arena[off:off + 8]
        res = compare_host_arenas(env.cfg, dev, ora.arena, what=f"chain of {chain}", max_flip_frac=0.05 * chain, max_event_frac=2e-3 * chain * chain)
        tally.add(res)
    print(tally.line(f"chained teacher n={n} ({nsteps} steps in 6 chains)"))
    assert tally.events <= 5e-4 * n * nsteps  # observed: 5 in 10 steps


def test_lost_chain_announcement_is_an_error_not_a_hang():
    """The consumers of a chained launch poll a bounded number of times for the announcement of their step's command block.  With
    the publisher announcing a wrong step id once (lt_env_defer_gate mode 3, a test hook) the launch still ends, lt_env_check
    reports LT_EHIP, and the env works again afterwards."""
    import torch

    n = 256
    env = make_env("teacher", n)
    act = torch.zeros(n, 12, device="cuda:0")
    env.defer_gate(2)
    env.step_rows_raw(act.data_ptr(), 0, 0, 0, 0)
    env.check()  # a healthy chain raises nothing
    env.defer_gate(3)
    env.step_rows_raw(act.data_ptr(), 0, 0, 0, 0)  # its publisher announces step + 1: every other workgroup times out (~50 ms)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="lt_env_check"):
        env.check()
    env.gate_update()
    env.defer_gate(0)
    env.check()  # cleared
    obs, rew, dones, _ = env.step(act)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_chained_population_pass_equals_the_pass_behind_every_step():
    """lt_env_defer_gate mode 2: the population pass of step t (curriculum decision, population gate, step counter) runs inside the
    launch of step t + 1, on workgroup 0's RNG wave beside the physics; the other workgroups read the command block it leaves
    through the flag / sc1 protocol.  Chains of irregular length against the default mode on a twin env: identical arenas, bit for
    bit (counters[2], the chain's scratch flag, aside), with a curriculum that really widens the command ranges on the way."""
    import torch
    from locotouch_amd.env import LocoTouchVecEnv

    n = 4096  # 256 tiles: every CU has one, so the cross-workgroup hand-off of the command block is exercised across all XCDs
    cfg = _abi.preset_cfg(TASKS["teacher"], num_envs=n, seed=23)
    cfg.max_episode_length, cfg.cur_len_threshold = 10, 2.0  # frequent time-outs: the gate opens and the ranges widen within the run
    cfg.cur_reward_threshold[0] = cfg.cur_reward_threshold[1] = -1.0e3
    a = LocoTouchVecEnv(TASKS["teacher"], device="cuda:0", cfg=cfg)
    b = LocoTouchVecEnv(TASKS["teacher"], device="cuda:0", cfg=cfg)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    P0 = a.cmd_params.clone()
    t = 0
    for chain in (1, 2, 5, 24, 3, 24, 7):
        a.defer_gate(2)
        for _ in range(chain):
            act = 0.4 * torch.randn(n, 12, device="cuda:0", generator=g)
            a.step_rows_raw(act.data_ptr(), 0, 0, 0, 0)
            b.step(act)
            t += 1
        a.gate_update()
        a.defer_gate(0)
        torch.cuda.synchronize()
        ha, hb = a._arena_aligned.clone(), b._arena_aligned.clone()
        L = Layout(n, a.num_obs)
        off = L.plain["LT_F_COUNTERS"][0] + 16
        ha[off:off + 8] = 0
        hb[off:off + 8] = 0
        assert torch.equal(ha, hb), f"after {t} steps (chain of {chain})"
        assert int(a.counters[0]) == t + 1
    assert not torch.equal(a.cmd_params[:6], P0[:6]), "the command ranges must have widened"
    # a pass left outstanding is absorbed by the next launch whatever the mode; mode 0 cannot be selected over it
    a.defer_gate(1)
    a.step_rows_raw(act.data_ptr(), 0, 0, 0, 0)
    with pytest.raises(RuntimeError):
        a.defer_gate(0)
    a.gate_update()
    a.defer_gate(0)


@pytest.mark.parametrize("task,n", [("locomotion", 48), ("teacher", 8208)])
def test_chained_mode_on_other_grids(task, n):
    """Chained mode on a locomotion env (no object: the helper waves' object roles are idle) and on a grid beyond one tile per CU,
    where the single-wave form cannot absorb a pass and the launcher issues it itself: identical arenas to the default mode."""
    import torch

    a, b = make_env(task, n, seed=5), make_env(task, n, seed=5)
    g = torch.Generator(device="cuda:0").manual_seed(2)
    L = Layout(n, a.num_obs)
    off = L.plain["LT_F_COUNTERS"][0] + 16
    for chain in (3, 1, 6):
        a.defer_gate(2)
        for _ in range(chain):
            act = 0.4 * torch.randn(n, 12, device="cuda:0", generator=g)
            a.step_rows_raw(act.data_ptr(), 0, 0, 0, 0)
            b.step(act)
        a.gate_update()
        a.defer_gate(0)
        torch.cuda.synchronize()
        ha, hb = a._arena_aligned.clone(), b._arena_aligned.clone()
        ha[off:off + 8] = 0
        hb[off:off + 8] = 0
        assert torch.equal(ha, hb), f"{task} n={n} after a chain of {chain}"
    assert int(a.counters[0]) == 11


def test_free_running_statistics_teacher():
    """Without re-syncing, chaotic contact dynamics decorrelate trajectories; episode statistics must still agree."""
    import torch

    n, steps = 256, 200
    env = make_env("teacher", n)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    L = Layout(n, env.num_obs)
    g = torch.Generator().manual_seed(5)
    dev_rew, ora_rew, dev_done, ora_done = 0.0, 0.0, 0, 0
    for t in range(steps):
        act = 0.3 * torch.randn(n, 12, generator=g)
        _, rew, dones, _ = env.step(act.cuda())
        ora.step(act.numpy(), nthreads=8)
        dev_rew += float(rew.sum()); dev_done += int(dones.sum())
        ora_rew += float(L.arr(ora.arena, "LT_F_REWARD")[:n].sum()); ora_done += int(L.arr(ora.arena, "LT_F_DONES")[:n].sum())
    assert abs(dev_done - ora_done) <= 0.15 * max(ora_done, 20), (dev_done, ora_done)
    assert abs(dev_rew - ora_rew) <= 0.15 * abs(ora_rew) + 5.0, (dev_rew, ora_rew)


# ------------------------------------------------------------------------------------------------------------------
# HIP path against the reference's golden vectors (term level): lt_env_eval_terms on golden states
# ------------------------------------------------------------------------------------------------------------------
def _write_golden_state(env, L, g, t, hist_prev):
    """Golden record t (IsaacLab data-contract layout) -> arena fields (host), returns the host arena."""
    n = env.num_envs
    a = device_arena_to_host(env)
    z = np.zeros((n, 1), np.float32)
    L.set_vec(a, "LT_F_ROOT_POS", np.concatenate([g["robot_root_pos_w"][t], z], 1))
    L.set_vec(a, "LT_F_ROOT_QUAT", g["robot_root_quat_w"][t])
    L.set_vec(a, "LT_F_ROOT_LIN_VEL_W", np.concatenate([g["robot_root_lin_vel_w"][t], z], 1))
    L.set_vec(a, "LT_F_ROOT_ANG_VEL_W", np.concatenate([g["robot_root_ang_vel_w"][t], z], 1))
    L.set_vec(a, "LT_F_JOINT_POS", g["robot_joint_pos"][t])
    L.set_vec(a, "LT_F_JOINT_VEL", g["robot_joint_vel"][t])
    L.set_vec(a, "LT_F_JOINT_ACC", g["robot_joint_acc"][t])
    L.set_vec(a, "LT_F_APPLIED_TORQUE", g["robot_applied_torque"][t])
    L.set_vec(a, "LT_F_ACT_RAW", g["raw_actions"][t])
    L.set_vec(a, "LT_F_ACT_PREV_RAW", g["prev_raw_actions"][t])
    norms = np.linalg.norm(g["net_forces_w"][t].astype(np.float32), axis=-1).astype(np.float32)  # (n,17)
    hist = [norms] + hist_prev[:2]
    fh = np.zeros((n, 48), np.float32)
    for s in range(3):
        for ty in range(4):
            fh[:, (s * 4 + ty) * 4:(s * 4 + ty) * 4 + 4] = hist[s][:, 1 + ty * 4:1 + ty * 4 + 4]
    L.set_vec(a, "LT_F_FORCE_HIST", fh)
    L.set_vec(a, "LT_F_TRUNK_FORCE_HIST", np.stack([hist[0][:, 0], hist[1][:, 0], hist[2][:, 0], z[:, 0]], 1))
    L.set_vec(a, "LT_F_FOOT_CUR_AIR", g["current_air_time"][t])
    L.set_vec(a, "LT_F_FOOT_CUR_CONTACT", g["current_contact_time"][t])
    L.set_vec(a, "LT_F_FOOT_LAST_AIR", g["last_air_time"][t])
    L.set_vec(a, "LT_F_FOOT_LAST_CONTACT", g["last_contact_time"][t])
    fp, fv = g["robot_body_pos_w"][t], g["robot_body_lin_vel_w"][t]  # (n,4,3)
    L.set_vec(a, "LT_F_FOOT_POS_W", fp.transpose(0, 2, 1).reshape(n, 12))
    L.set_vec(a, "LT_F_FOOT_VEL_W", fv.transpose(0, 2, 1).reshape(n, 12))
    L.set_vec(a, "LT_F_OBJ_POS", np.concatenate([g["obj_root_pos_w"][t], z], 1))
    L.set_vec(a, "LT_F_OBJ_QUAT", g["obj_root_quat_w"][t])
    L.set_vec(a, "LT_F_OBJ_LIN_VEL_W", np.concatenate([g["obj_root_lin_vel_w"][t], z], 1))
    L.set_vec(a, "LT_F_OBJ_ANG_VEL_W", np.concatenate([g["obj_root_ang_vel_w"][t], z], 1))
    L.set_vec(a, "LT_F_OBJ_TIMERS", np.concatenate([g["obj_current_air_time"][t], g["obj_current_contact_time"][t],
                                                   g["obj_last_air_time"][t], g["obj_last_contact_time"][t]], 1))
    cmd = L.vec(a, "LT_F_CMD")
    cmd[:, :3] = g["cmd"][t]
    L.set_vec(a, "LT_F_CMD", cmd)
    L.arr(a, "LT_F_TERMINATED")[:n] = g["terminated"][t].astype(np.uint8)
    # class-term resets before this step's call
    rs = g["gait_reset"][t]
    for name in ("LT_F_GAIT_LAST_AIR", "LT_F_GAIT_LAST_CONTACT", "LT_F_GAIT_VALID_LAST_AIR", "LT_F_GAIT_FLAGS", "LT_F_GAIT_CMD"):
        v = L.vec(a, name)
        v[rs] = 0
        L.set_vec(a, name, v)
    return a, hist


def test_reward_terms_against_reference_golden():
    """The HIP reward/termination/object-observation code against what the reference's own functions returned
    (tests/golden/mdp_rewards_teacher.npz), including the stateful gait class across the 200-step sequence."""
    import torch
    from tests.test_oracle_golden import TERM_MAP

    g = dict(np.load(os.path.join(GOLD, "mdp_rewards_teacher.npz")))
    T, n = g["cmd"].shape[:2]
    env = make_env("teacher", n, enable_corruption=0)
    cfg = env.cfg
    for i in range(C["LT_NUM_REWARD_TERMS"]):
        if cfg.reward_weight[i] == 0:
            cfg.reward_weight[i] = 1.0
    # rebuild the env handle with the modified weights (cfg is captured at create time)
    from locotouch_amd.env import LocoTouchVecEnv
    env = LocoTouchVecEnv(TASKS["teacher"], num_envs=n, device="cuda:0", cfg=cfg)
    L = Layout(n, env.num_obs)
    hist = [np.zeros((n, 17), np.float32)] * 2
    worst = {}
    for t in range(T):
        a, hist = _write_golden_state(env, L, g, t, hist)
        env._arena_aligned.copy_(torch.from_numpy(a))
        env.eval_terms()
        torch.cuda.synchronize()
        out = device_arena_to_host(env)
        terms = L.vec(out, "LT_F_REWARD_TERMS")
        for name, key in list(TERM_MAP.items()) + [("gait_with_object", "LT_R_GAIT")]:
            ref = g["out_" + name][t].astype(np.float32)
            err = np.abs(terms[:, C[key]] - ref) / np.maximum(1.0, np.abs(ref))
            worst[name] = max(worst.get(name, 0.0), float(err.max()))
        bits = L.arr(out, "LT_F_TERM_BITS")[:n]
        assert ((bits >> C["LT_T_OBJECT_BELOW_ROBOT"]) & 1 == g["out_term_object_below_robot"][t]).all()
        assert ((bits >> C["LT_T_OBJECT_BAD_ROLL"]) & 1 == g["out_term_object_bad_roll"][t]).all()
        obs = L.arr(out, "LT_F_OBS_CRITIC")[:n]
        np.testing.assert_allclose(obs[:, 270 + 65:270 + 78], g["out_obs_object_state"][t], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(L.vec(out, "LT_F_GAIT_VALID_LAST_AIR")[:, [0, 3, 1, 2]], g["gait_state_valid_last_air_time"][t], atol=1e-6)
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, f"HIP terms off the reference golden beyond 1e-4 (rel): {bad}"


# ------------------------------------------------------------------------------------------------------------------
# full size (BASELINE.json: 4096 envs/GPU): size-independent properties
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("task,n", [("teacher", 4096), ("locomotion", 4096), ("teacher", 32768), ("locomotion", 32768)])
def test_full_size_properties(task, n):
    """n = 4096: BASELINE.json's headline size (LDS-DMA history path); n = 32768: config 5's size (one-wave form: history rows staged one group at a time)."""
    import torch

    env = make_env(task, n, seed=42)
    env2 = make_env(task, n, seed=42)
    g = torch.Generator().manual_seed(1)
    prev_obs = env.obs_policy.clone()
    prev_len = env.episode_length_buf.clone()
    dims = [3, 3, 3, 12, 12, 12] + ([13] if task == "teacher" else [])
    total_done = 0
    for t in range(60):
        act = (0.5 * torch.randn(n, 12, generator=g)).cuda()
        obs, rew, dones, extras = env.step(act)
        env2.step(act)
        torch.cuda.synchronize()
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(env.obs_critic).all()
        d = dones.bool()
        # reset / episode indexing is integer-exact: ep_len increments by one, or is zero right after a reset
        assert torch.equal(env.episode_length_buf[~d], prev_len[~d] + 1) and (env.episode_length_buf[d] == 0).all()
        assert torch.equal(dones, (env.terminated_buf | env.time_out_buf).long())
        # history: slot s of step t == slot s+1 of step t-1 for envs that did not reset; all slots equal after a reset
        off = 0
        for dd in dims:
            cur = obs[:, off:off + 6 * dd].reshape(n, 6, dd)
            old = prev_obs[:, off:off + 6 * dd].reshape(n, 6, dd)
            assert torch.equal(cur[~d][:, :5], old[~d][:, 1:])
            assert (cur[d] == cur[d][:, -1:, :]).all()
            off += 6 * dd
        prev_obs = obs.clone(); prev_len = env.episode_length_buf.clone()
        total_done += int(d.sum())
    # determinism: same seed, same actions -> byte-identical arenas (counter-based RNG, no atomics)
    assert torch.equal(env._arena_aligned, env2._arena_aligned)
    assert total_done > 0
    assert int(env.counters[0]) == 61


def test_time_out_fires_exactly_at_max_episode_length():
    """time_out is an integer compare on episode_length_buf (bit-exact requirement)."""
    import torch

    n = 64
    env = make_env("teacher", n)
    env.episode_length_buf = torch.full((n,), 997, dtype=torch.long, device="cuda:0")
    fired, term = [], []
    for t in range(4):
        _, _, dones, extras = env.step(torch.zeros(n, 12, device="cuda:0"))
        torch.cuda.synchronize()
        fired.append(extras["time_outs"].clone())
        term.append(env.terminated_buf.bool().clone())
        if t == 2:
            assert (env.episode_length_buf[fired[2]] == 0).all()
    # ep_len runs 998, 999, 1000 -> time_out on the third step in exactly the envs that were not reset (terminated) before it
    assert not fired[0].any() and not fired[1].any() and not fired[3].any()
    expected = ~(term[0] | term[1])
    assert expected.any() and torch.equal(fired[2], expected)
    assert (env.episode_length_buf <= env.max_episode_length).all()


def test_velocity_curriculum_kernel_matches_reference_golden():
    """The curriculum pass of the step kernel's tail (per-wave partials, last arriver decides) against the reference's own
    curriculum sequence (tests/golden/mdp_curriculum.npz): lt_env_curriculum_update feeds the golden reset batches as the
    per-env records the step kernel would derive.  The oracle twin runs beside it: byte-equal command block and trackers."""
    import torch

    g = np.load(os.path.join(GOLD, "mdp_curriculum.npz"))
    calls, n = g["reset_mask"].shape
    env = make_env("teacher", n)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    P = env.cmd_params
    L = Layout(n, env.num_obs)
    for c in range(calls):
        m = torch.from_numpy(g["reset_mask"][c])
        r = torch.zeros(n, 4)
        r[m, 0] = 1.0
        r[m, 1] = torch.from_numpy(g["ep_len"][c]).float()[m]
        r[m, 2] = torch.from_numpy(g["sum_lin"][c])[m]
        r[m, 3] = torch.from_numpy(g["sum_ang"][c])[m]
        env.curriculum_update(r.cuda())
        ora.curriculum_update(r.numpy())
        p = P.cpu().numpy()
        ref = g["ranges"][c]
        np.testing.assert_allclose(p[0:6].reshape(3, 2), ref[0:3], rtol=0, atol=2e-6, err_msg=f"call {c}")
        np.testing.assert_allclose(p[6:12].reshape(3, 2), ref[3:6], rtol=0, atol=2e-6, err_msg=f"call {c}")
        assert [bool(x) for x in p[12:15]] == [bool(x) for x in g["equal"][c]], f"call {c}"
        assert int(p[15]) == int(g["zero_steps"][c]) and abs(p[16] - float(g["rel_standing"][c])) < 1e-7
        assert int(p[17]) == int(g["lin_bins"][c]) and int(p[18]) == int(g["ang_bins"][c]), f"call {c}"
        a = device_arena_to_host(env)
        np.testing.assert_array_equal(L.arr(a, "LT_F_CMD_PARAMS"), L.arr(ora.arena, "LT_F_CMD_PARAMS"), err_msg=f"call {c}")
        np.testing.assert_array_equal(L.vec(a, "LT_F_CURRICULUM"), L.vec(ora.arena, "LT_F_CURRICULUM"), err_msg=f"call {c}")
    assert int(p[17]) > 5 and int(p[18]) > 5
    assert int(env.counters[0]) == 1 and int(env.counters[2]) == 0  # no step-counter increment; arrival ticket re-armed


def test_fused_rollout_kernels_match_torch():
    """lt_rollout_act / lt_rollout_record against the torch formulas they replace (ppo.py:129-170, rollout_storage.py:79-107)."""
    import torch
    from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
    from tests.rl_synth import POLICY_CFG, PPO_CFG

    n = 4096
    env = make_env("teacher", n)
    torch.manual_seed(3)
    alg = PPO(ActorCritic(348, 348, 12, **POLICY_CFG), device="cuda:0", **PPO_CFG)
    with torch.no_grad():
        alg.actor_critic.std.copy_(torch.linspace(0.3, 1.4, 12))
    alg.init_storage(n, 4, [348], [348], [12])
    fr = FusedRollout(env, alg)
    obs0, cobs0 = env.obs_policy.clone(), env.obs_critic.clone()
    with torch.inference_mode():
        mu_ref = alg.actor_critic.actor(obs0)
        v_ref = alg.actor_critic.critic(cobs0)
    fr.rollout(3)
    torch.cuda.synchronize()
    st = alg.storage
    # rows written straight into the storage slots == the in-place path of lt_env_step on a twin env, bit for bit
    twin = make_env("teacher", n)
    for t in range(3):
        o, r, d, _ = twin.step(st.actions[t].clone())
        nxt_p = st.observations[t + 1] if t < 2 else env.obs_policy
        nxt_c = st.privileged_observations[t + 1] if t < 2 else env.obs_critic
        assert torch.equal(o, nxt_p) and torch.equal(twin.obs_critic, nxt_c), f"step {t}"
        assert torch.equal(d.to(torch.uint8), st.dones[t].squeeze(1))
    # (counters[2] is scratch: the chained rollout leaves its last step id there, include/lt_env.h)
    assert torch.equal(twin.counters[[0, 3]], env.counters[[0, 3]]) and torch.equal(twin.cmd_params, env.cmd_params)
    assert torch.equal(st.observations[0], obs0) and torch.equal(st.privileged_observations[0], cobs0)
    torch.testing.assert_close(st.mu[0], mu_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(st.values[0], v_ref, rtol=1e-5, atol=1e-5)
    std = alg.actor_critic.std.detach()
    assert torch.equal(st.sigma[0], std.expand(n, 12))
    z = (st.actions[0] - st.mu[0]) / std
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02 and abs(float((z ** 4).mean()) - 3.0) < 0.15
    lp = torch.distributions.Normal(st.mu[0], std.expand(n, 12)).log_prob(st.actions[0]).sum(-1, keepdim=True)
    torch.testing.assert_close(st.actions_log_prob[0], lp, rtol=1e-4, atol=1e-4)
    assert not torch.equal(st.actions[0], st.actions[1])  # fresh noise every step (device-resident step counter)
    # record kernel: slot 2 holds the transition of the last env step
    exp_rew = env.reward_buf + alg.gamma * st.values[2].squeeze(1) * env.time_out_buf.float()
    torch.testing.assert_close(st.rewards[2].squeeze(1), exp_rew, rtol=1e-6, atol=1e-6)
    assert torch.equal(st.dones[2].squeeze(1), env.dones_buf.to(torch.uint8))
    # the fused rollout is hipGraph-capturable and the update consumes its storage unchanged
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fr.rollout(4)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        fr.rollout(4)
    g.replay()
    torch.cuda.synchronize()
    alg.compute_returns(env.obs_critic)
    losses = alg.update()
    assert all(np.isfinite(x) for x in losses[:3])


def test_rollout_rows_large_grid_register_history_path():
    """> 8192 envs: the one-wave form's history pass (old rows staged in LDS one group at a time) writing rows into
    rollout-storage slots (distinct prev/next pointers) must equal, bit for bit, the same pass shifting the arena rows in place (which test_step_parity_resynced pins to
    the oracle at this size)."""
    import torch
    from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
    from tests.rl_synth import POLICY_CFG, PPO_CFG

    n = 8208
    env, twin = make_env("teacher", n), make_env("teacher", n)
    torch.manual_seed(3)
    alg = PPO(ActorCritic(348, 348, 12, **POLICY_CFG), device="cuda:0", **PPO_CFG)
    alg.init_storage(n, 6, [348], [348], [12])
    fr = FusedRollout(env, alg)
    assert fr.rows_in_storage
    fr.rollout(6)
    torch.cuda.synchronize()
    st = alg.storage
    shifted = 0
    for t in range(6):
        o, r, d, _ = twin.step(st.actions[t].clone())
        nxt_p = st.observations[t + 1] if t < 5 else env.obs_policy
        nxt_c = st.privileged_observations[t + 1] if t < 5 else env.obs_critic
        assert torch.equal(o, nxt_p) and torch.equal(twin.obs_critic, nxt_c), f"step {t}"
        assert torch.equal(d.to(torch.uint8), st.dones[t].squeeze(1))
        shifted += int((d == 0).sum())
    assert shifted > 5 * n
    assert torch.equal(twin._arena_aligned, env._arena_aligned)


def test_training_loop_runs_and_checkpoints(tmp_path):
    """OnPolicyRunner end to end on the HIP env (fused rollout + PPO update): finite losses, checkpoint round trip."""
    import torch
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    task = TASKS["teacher"]
    env = make(task, num_envs=512, device="cuda:0", seed=1)
    cfg = train_cfg(task)
    runner = OnPolicyRunner(env, cfg, log_dir=str(tmp_path), device="cuda:0")
    runner.learn(3, init_at_random_ep_len=True)
    h = runner.history
    assert len(h) == 3 and all(np.isfinite(r["Loss/value_function"]) and np.isfinite(r["Loss/surrogate"]) for r in h)
    assert h[-1]["Perf/total_fps"] > 0 and "Metrics/base_velocity/lin_vel_x" in h[-1]
    ck = os.path.join(str(tmp_path), "model_2.pt")  # iterations 0..2: the final save carries the last iteration's number
    assert os.path.exists(ck) and "Metrics/base_velocity/error_vel_xy" in h[-1]
    loaded = torch.load(ck, weights_only=True)
    assert set(loaded) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"}  # on_policy_runner.py:369-385
    runner2 = OnPolicyRunner(make(task, num_envs=512, device="cuda:0", seed=2), cfg, log_dir=None, device="cuda:0")
    runner2.load(ck)
    for a, b in zip(runner.alg.actor_critic.parameters(), runner2.alg.actor_critic.parameters()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dims,act,m", [((348, 512, 256, 128, 12), "elu", 4096), ((348, 512, 256, 128, 1), "elu", 1000),
                                        ((45, 64, 64, 7), "relu", 37), ((270, 256, 128, 128, 12), "tanh", 50), ((33, 20), "elu", 16),
                                        # the locomotion task's width (2 mod 4: pair-wise input staging) at two and four row tiles, and inputs
                                        # wider than one staging batch on the pair-wise and the float4 path
                                        ((270, 512, 256, 128, 12), "elu", 9000), ((270, 512, 256, 128, 1), "elu", 40000),
                                        ((774, 64, 8), "elu", 300), ((1000, 40, 4), "elu", 100)])
def test_fused_mlp_matches_torch(dims, act, m):
    """lt_mlp_forward (f16 MFMA with error-compensated operand splitting, one launch) against the torch fp32 modules it replaces; tolerance 2e-5 * scale
    (both are fp32 accumulations, only the summation order differs)."""
    import torch
    from locotouch_amd.rl.mlp import PackedMLP
    from locotouch_amd.rl.modules import build_mlp

    torch.manual_seed(5)
    seq = build_mlp(dims[0], list(dims[1:-1]), dims[-1], act).to("cuda:0")
    with torch.no_grad():
        for p in seq.parameters():
            p.mul_(2.0)  # livelier activations than the default init
    x = torch.randn(m, dims[0], device="cuda:0") * 1.5
    with torch.inference_mode():
        ref = seq(x)
        ref64 = seq.double()(x.double())
    seq.float()
    net = PackedMLP(seq)
    y = net(x)
    torch.cuda.synchronize()
    scale = float(ref64.abs().max()) + 1.0
    err = float((y.double() - ref64).abs().max())
    err_torch = float((ref.double() - ref64).abs().max())
    assert err < 2e-5 * scale, (err, err_torch)
    assert err < 4 * err_torch + 1e-6 * scale  # no worse than torch's own fp32 error by more than a small factor
    # parameters change -> pack() picks them up
    with torch.no_grad():
        for p in seq.parameters():
            p.add_(0.01)
    net.pack()
    with torch.inference_mode():
        torch.testing.assert_close(net(x), seq(x), rtol=2e-5, atol=2e-5 * scale)


def test_fused_mlp_saturates_layer_inputs_at_the_documented_bound():
    """include/lt_env.h, LT_MLP_INPUT_CLAMP: every value entering a layer (input rows, hidden activations) is saturated to +-1000 before
    the f16 operand split; the kernel then equals the fp32 stack evaluated on clamped layer inputs - pinned here just above the
    bound and far above it - and the trainer-side monitor (PackedPair.domain_violated) sees it."""
    import torch
    from locotouch_amd import _abi
    from locotouch_amd.rl.mlp import PackedMLP, PackedPair
    from locotouch_amd.rl.modules import build_mlp

    B = float(_abi.CONSTS["LT_MLP_INPUT_CLAMP"])
    assert B == 1000.0
    torch.manual_seed(9)
    seq = build_mlp(64, [128, 64], 12, "elu").to("cuda:0")
    with torch.no_grad():
        seq[0].weight.mul_(6.0)  # hidden activations beyond the bound for the large inputs below
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)]

    def clamped_stack(x):
        h = x
        for i, l in enumerate(lin):
            h = l(h.clamp(-B, B))
            if i < len(lin) - 1:
                h = torch.nn.functional.elu(h)
        return h

    net = PackedMLP(seq)
    x = torch.randn(256, 64, device="cuda:0")
    x[:64] *= 1.0          # inside the domain: equal to the plain stack
    x[64:128] = x[64:128].sign() * (B + 0.5 + 3.0 * torch.rand(64, 64, device="cuda:0"))  # just above the bound
    x[128:192] *= 20000.0  # far above (still finite in f16 before the clamp would matter: up to ~6e4)
    x[192:] *= 300.0       # inputs inside, first hidden layer beyond the bound
    with torch.inference_mode():
        y, want, plain = net(x), clamped_stack(x), seq(x)
    scale = float(want.abs().max()) + 1.0
    torch.testing.assert_close(y, want, rtol=3e-5, atol=3e-5 * scale)
    torch.testing.assert_close(y[:64], plain[:64], rtol=3e-5, atol=3e-5 * scale)
    assert float((plain[64:] - want[64:]).abs().max()) > 1e-2 * scale, "the case must actually leave the domain"
    assert bool(torch.isfinite(y).all())
    # the monitor: armed once, it records the largest layer input of the next training forward
    critic = build_mlp(64, [128, 64], 1, "elu").to("cuda:0")
    pair = PackedPair(seq, critic)
    pair.arm_domain_check()
    pair(x[:64].contiguous(), x[:64].contiguous())
    assert not pair.domain_violated()
    pair.arm_domain_check()
    pair(x[64:128].contiguous(), x[:64].contiguous())
    assert pair.domain_violated() and not pair.domain_violated()  # (reading resets the record)


def test_fused_policy_kernel_matches_act_kernel():
    """lt_rollout_policy (actor MLP + sampling in one launch) against actor GEMMs + lt_rollout_act on a twin env."""
    import torch
    from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
    from tests.rl_synth import POLICY_CFG, PPO_CFG

    n = 1024
    outs = []
    for packed in (True, False):
        env = make_env("teacher", n)
        torch.manual_seed(3)
        alg = PPO(ActorCritic(348, 348, 12, **POLICY_CFG), device="cuda:0", **PPO_CFG)
        alg.init_storage(n, 3, [348], [348], [12])
        fr = FusedRollout(env, alg, use_packed_mlp=packed)
        assert (fr.actor_mlp is not None) == packed
        fr.rollout(3)
        torch.cuda.synchronize()
        outs.append(alg.storage)
    a, b = outs
    # step 0 sees identical observations; later steps diverge only through ~1e-6 action differences
    torch.testing.assert_close(a.mu[0], b.mu[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.actions[0], b.actions[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.actions_log_prob[0], b.actions_log_prob[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a.values[0], b.values[0], rtol=1e-5, atol=1e-5)
    assert torch.equal(a.sigma[0], b.sigma[0])
    assert not torch.equal(a.actions[1] - a.mu[1], a.actions[0] - a.mu[0])  # fresh noise per step (counter + t keying)
    z_a, z_b = (a.actions[2] - a.mu[2]), (b.actions[2] - b.mu[2])
    torch.testing.assert_close(z_a, z_b, rtol=1e-4, atol=1e-5)  # same Philox draws at step 2 in both paths


def test_back_to_back_graph_replays_of_the_rollout_equal_eager_rollouts():
    """bench.py and the trainer replay the captured 24-step rollout graph several times in a row with nothing in between.
    Replays must be ordered by the stream (replay k+1 reads the arena replay k wrote): three back-to-back replays of a 6-step
    rollout graph must leave the env and the storage exactly where 18 eagerly launched steps leave a twin."""
    import copy

    import torch
    from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
    from tests.rl_synth import POLICY_CFG, PPO_CFG

    n, T = 4096, 6
    envs = [make_env("teacher", n), make_env("teacher", n)]
    torch.manual_seed(3)
    ac = ActorCritic(348, 348, 12, **POLICY_CFG)
    algs = [PPO(copy.deepcopy(ac), device="cuda:0", **PPO_CFG) for _ in range(2)]
    frs = []
    for env, alg in zip(envs, algs):
        alg.init_storage(n, T, [348], [348], [12])
        frs.append(FusedRollout(env, alg))
    # A: capture once (warm-up on a side stream advances the env: the twin does the same rollout eagerly), then 3 replays
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        frs[0].rollout(T)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        frs[0].rollout(T)
    for _ in range(3):
        g.replay()
    # B: the same 4 rollouts, launched eagerly
    for _ in range(4):
        frs[1].rollout(T)
    torch.cuda.synchronize()
    assert torch.equal(envs[0].counters[:1], envs[1].counters[:1]) and int(envs[0].counters[0]) == 1 + 4 * T
    assert torch.equal(envs[0]._arena_aligned, envs[1]._arena_aligned), "arena differs: replays were not ordered"
    for name in ("observations", "privileged_observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu"):
        assert torch.equal(getattr(algs[0].storage, name), getattr(algs[1].storage, name)), name
