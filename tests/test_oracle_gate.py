"""Multi-rank curriculum gate, CPU: the decision sequence on population SUMS must be the per-env
`lt_oracle_curriculum` (itself pinned to the reference's ModifyVelCommandsRangeBasedonReward by tests/golden/mdp_curriculum.npz)
on the same population; R shards with `env_index_offset` reproduce one big population draw for draw; and a 2-rank gloo run
widens the command ranges on the same step on both ranks - the step a single process with the whole population picks."""
import ctypes
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from locotouch_amd import _abi
from locotouch_amd.layout import Layout
from tests import oracle_lib as O

C = _abi.CONSTS
TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"


def _P(cfg):
    P = np.zeros(C["LT_CMD_PARAMS_LEN"], np.float32)
    O.load().lt_oracle_cmd_params_init(ctypes.byref(cfg), O.fptr(P))
    return P


def test_gate_on_sums_is_the_per_env_curriculum():
    lib = O.load()
    cfg = _abi.preset_cfg(TASK, num_envs=48)
    rng = np.random.default_rng(0)
    n = 48
    widened = 0
    for trial in range(300):
        P = _P(cfg)
        for _ in range(int(rng.integers(0, 4))):  # some history: earlier widenings of either group
            P[17 + int(rng.integers(0, 2))] += 1
        trk = np.zeros((n, 8), np.float32)
        frac = rng.choice([0.0, 0.5, 1.0])
        had = rng.random(n) < frac
        trk[had, 0] = trk[had, 3] = 1
        trk[had, 1] = trk[had, 4] = rng.uniform(15, 40, had.sum()).astype(np.float32)
        trk[had, 2] = rng.uniform(10, 25, had.sum()).astype(np.float32)
        trk[had, 5] = rng.uniform(4, 12, had.sum()).astype(np.float32)
        rec = np.zeros((n, 4), np.float32)
        rs = rng.random(n) < rng.choice([0.0, 0.2, 1.0])
        if frac == 1.0 or rng.random() < 0.5:
            rs |= ~had  # everybody has reset at least once after this pass
        rec[rs, 0] = 1
        rec[rs, 1] = rng.uniform(15, 40, rs.sum()).astype(np.float32)
        rec[rs, 2] = rng.uniform(10, 25, rs.sum()).astype(np.float32)
        rec[rs, 3] = rng.uniform(4, 12, rs.sum()).astype(np.float32)
        # population sums with this pass's records merged, per 16-env group as the kernel's waves form them
        m_len_l = np.where(rs, rec[:, 1], trk[:, 1]); m_sum_l = np.where(rs, rec[:, 2], trk[:, 2])
        m_len_a = np.where(rs, rec[:, 1], trk[:, 4]); m_sum_a = np.where(rs, rec[:, 3], trk[:, 5])
        grp = lambda v: v.reshape(-1, 16).any(axis=1).sum()  # noqa: E731
        r = np.array([3, grp(rs), grp(~(rs | (trk[:, 0] != 0))), m_len_l.sum(), m_sum_l.sum(), grp(~(rs | (trk[:, 3] != 0))),
                      m_len_a.sum(), m_sum_a.sum()], np.float32)
        P_a, P_b = P.copy(), P.copy()
        trk_a = trk.copy()
        lib.lt_oracle_curriculum(ctypes.byref(cfg), O.fptr(P_a), n, O.fptr(rec), O.fptr(trk_a))
        out = (ctypes.c_int32 * 5)()
        lib.lt_oracle_gate_on_sums(ctypes.byref(cfg), O.fptr(P_b), O.fptr(r), np.float32(1.0 / n), 1, 1, out)
        np.testing.assert_array_equal(P_a[:24], P_b[:24])
        widened += int(P_b[17] + P_b[18] > P[17] + P[18])
        if out[2]:
            assert (trk_a[:, :3] == 0).all()
        if out[4]:
            assert (trk_a[:, 3:6] == 0).all()
    assert 20 < widened < 280


def test_shards_with_env_index_offset_reproduce_one_population():
    """Two 32-env shards (offsets 0, 32; same seed) == envs [0,32) and [32,64) of one 64-env population, bit for bit."""
    big = _abi.preset_cfg(TASK, num_envs=64, seed=9)
    ob = O.OracleEnv(big)
    ob.reset_all()
    shards = []
    for r in range(2):
        c = _abi.preset_cfg(TASK, num_envs=32, seed=9)
        c.env_index_offset = 32 * r
        o = O.OracleEnv(c)
        o.reset_all()
        shards.append(o)
    g = np.random.default_rng(1)
    Lb, Ls = Layout(64, 348), Layout(32, 348)
    for t in range(40):
        a = (0.5 * g.standard_normal((64, 12))).astype(np.float32)
        ob.step(a)
        for r, o in enumerate(shards):
            o.step(a[32 * r:32 * (r + 1)])
            for name in ("LT_F_OBS_POLICY", "LT_F_OBS_CRITIC", "LT_F_REWARD", "LT_F_DONES"):
                np.testing.assert_array_equal(Ls.arr(o.arena, name)[:32], Lb.arr(ob.arena, name)[32 * r:32 * (r + 1)], err_msg=f"{name} step {t}")
    assert Lb.arr(ob.arena, "LT_F_DONES")[:64].sum() >= 0


# ---- 2-rank gloo: the gate decided on all-reduced sums ------------------------------------------------------------------------
N_RANK, ROLLOUT, ITERS = 16, 8, 40


def _gate_cfg(n, offset, external):
    c = _abi.preset_cfg(TASK, num_envs=n, seed=4)
    c.env_index_offset, c.cur_gate_external = offset, external
    c.max_episode_length, c.episode_length_s = 12, 0.24  # every env times out often: the gate sees whole populations quickly
    c.cur_len_threshold = 2.0                             # and passes on modest episodes (the sums, not the physics, are under test)
    c.cur_reward_threshold[0] = c.cur_reward_threshold[1] = -1.0e3
    return c


def _run_rank(env, dist, actions, log):
    from tests.oracle_vec_env import OracleVecEnv  # noqa: F401

    L = env.layout
    for it in range(ITERS):
        for t in range(ROLLOUT):
            env.o.step(actions[it * ROLLOUT + t])
        env.curriculum_sync(dist, ROLLOUT)
        log.append(L.arr(env.o.arena, "LT_F_CMD_PARAMS")[:24].copy())


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from locotouch_amd.rl import Dist
    from tests.oracle_vec_env import OracleVecEnv

    dist = Dist.from_env(backend="gloo")
    env = OracleVecEnv(TASK, cfg=_gate_cfg(N_RANK, rank * N_RANK, 1))
    acts = np.load(os.path.join(outdir, "actions.npy"))[:, rank * N_RANK:(rank + 1) * N_RANK]
    log = []
    _run_rank(env, dist, acts, log)
    np.save(os.path.join(outdir, f"P_rank{rank}.npy"), np.stack(log))
    dist.shutdown()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_widen_on_the_same_step_as_one_population(world):
    """2 ranks, and the node's 8: 8 x 16 envs with env_index_offset r * 16 == one 128-env population (curriculum ring, RNG keys)."""
    from locotouch_amd.rl import Dist
    from tests.oracle_vec_env import OracleVecEnv

    outdir = tempfile.mkdtemp()
    rng = np.random.default_rng(2)
    actions = (0.3 * rng.standard_normal((ITERS * ROLLOUT, world * N_RANK, 12))).astype(np.float32)
    np.save(os.path.join(outdir, "actions.npy"), actions)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, outdir), nprocs=world, join=True)
    P0 = np.load(os.path.join(outdir, "P_rank0.npy"))
    for r in range(1, world):
        np.testing.assert_array_equal(P0, np.load(os.path.join(outdir, f"P_rank{r}.npy")))  # identical command block on every rank after every rollout
    bins = P0[:, 17] + P0[:, 18]
    assert bins[-1] >= 3 and (np.diff(bins) >= 0).all(), bins  # the ranges did widen, several times
    # one process holding the whole 32-env population, same lagged gate: the same widenings at the same rollouts
    env = OracleVecEnv(TASK, cfg=_gate_cfg(world * N_RANK, 0, 1))
    log = []
    _run_rank(env, Dist(), actions, log)
    np.testing.assert_allclose(np.stack(log), P0, rtol=0, atol=0)
    # and the lag against the reference's own per-step gate (cur_gate_external = 0) is bounded by one rollout per widening
    env = OracleVecEnv(TASK, cfg=_gate_cfg(world * N_RANK, 0, 0))
    log1 = []
    _run_rank(env, Dist(), actions, log1)
    b1 = np.stack(log1)[:, 17] + np.stack(log1)[:, 18]
    assert b1[-1] >= bins[-1] and (b1 >= bins).all()


# ---- the trainer's hook: OnPolicyRunner.learn() calls env.curriculum_sync(dist, steps) after every rollout --------------------
def _runner_worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.rl import Dist, OnPolicyRunner
    from tests.oracle_vec_env import OracleVecEnv

    dist = Dist.from_env(backend="gloo")
    env = OracleVecEnv(TASK, cfg=_gate_cfg(N_RANK, rank * N_RANK, 1))
    calls = []
    sync = env.curriculum_sync
    env.curriculum_sync = lambda d, n: (calls.append((d.world_size, n)), sync(d, n))[1]
    cfg = train_cfg(TASK)
    cfg["num_steps_per_env"], cfg["algorithm"]["num_learning_epochs"], cfg["algorithm"]["num_mini_batches"] = 8, 1, 1
    torch.manual_seed(0)
    runner = OnPolicyRunner(env, cfg, log_dir=None, device="cpu", dist=dist)
    runner.learn(6)
    P = env.layout.arr(env.o.arena, "LT_F_CMD_PARAMS")[:24].copy()
    np.save(os.path.join(outdir, f"runner_P_rank{rank}.npy"), P)
    np.save(os.path.join(outdir, f"runner_calls_rank{rank}.npy"), np.array(calls))
    sd = torch.cat([p.detach().flatten() for p in runner.alg.actor_critic.parameters()])
    np.save(os.path.join(outdir, f"runner_w_rank{rank}.npy"), sd.numpy())
    dist.shutdown()


def test_runner_syncs_the_gate_after_every_rollout_on_all_ranks():
    outdir = tempfile.mkdtemp()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_runner_worker, args=(2, port, outdir), nprocs=2, join=True)
    P0, P1 = np.load(os.path.join(outdir, "runner_P_rank0.npy")), np.load(os.path.join(outdir, "runner_P_rank1.npy"))
    np.testing.assert_array_equal(P0, P1)
    assert P0[17] + P0[18] >= 1, "the ranges must have widened at least once in 48 steps"
    for r in range(2):
        assert np.load(os.path.join(outdir, f"runner_calls_rank{r}.npy")).tolist() == [[2, 8]] * 6
    np.testing.assert_array_equal(np.load(os.path.join(outdir, "runner_w_rank0.npy")), np.load(os.path.join(outdir, "runner_w_rank1.npy")))
