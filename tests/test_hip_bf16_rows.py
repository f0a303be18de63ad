"""BASELINE config 5: bf16 observation rows (rows, their 6-deep history, rollout-storage observations); all state stays f32.

The env kernel reads bf16 rows from one storage slot and writes the next (lt_env_set_row_format); a frame is rounded to nearest-even
once, when it enters a row, and carried bit for bit afterwards.  Checked against the ORACLE (f32 rows) from byte-identical state:
rows within bf16 rounding (2^-8 relative), everything else at the f32 bands of tests/parity_util.py; the carried frames bit-exact;
the policy kernel on bf16 rows == the policy kernel on the same values in f32; a 32768-env rollout through the fused path."""
import numpy as np
import pytest

from locotouch_amd import _abi
from locotouch_amd.layout import Layout
from tests import oracle_lib as O
from tests.parity_util import Tally, compare_host_arenas, device_arena_to_host

pytestmark = pytest.mark.gpu
TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"
DIMS = (3, 3, 3, 12, 12, 12, 13)  # observation terms of the teacher rows, 6 slots each


def _shift_ok(prev_u16, next_u16, keep):
    """history slots 0..4 of every term in `next` == slots 1..5 in `prev`, bit for bit, for the envs in `keep`"""
    off = 0
    for d in DIMS:
        assert np.array_equal(next_u16[keep, off:off + 5 * d], prev_u16[keep, off + d:off + 6 * d]), f"term at column {off}"
        off += 6 * d


@pytest.mark.parametrize("n", [4096, 8208])  # four-wave form (one tile per CU) / one-wave form
def test_bf16_rows_match_the_oracle_within_bf16_rounding(n):
    import torch
    from locotouch_amd.env import LocoTouchVecEnv

    env = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=11, debug_terms=1)
    ora = O.OracleEnv(env.cfg)
    ora.reset_all()
    g = torch.Generator().manual_seed(5)
    for _ in range(40):
        ora.step((0.6 * torch.randn(n, 12, generator=g)).numpy(), nthreads=8)
    L = Layout(n, env.num_obs)
    tally = Tally(n)
    env.set_row_format(torch.bfloat16)
    nxt = [torch.zeros(n, env.num_obs, dtype=torch.bfloat16, device="cuda:0") for _ in range(2)]
    n_reset = 0
    for t in range(10):
        env._arena_aligned.copy_(torch.from_numpy(ora.arena))
        prev = [torch.from_numpy(L.arr(ora.arena, f)[:n].copy()).cuda().to(torch.bfloat16) for f in ("LT_F_OBS_POLICY", "LT_F_OBS_CRITIC")]
        act = 0.6 * torch.randn(n, 12, generator=g)
        a_dev = act.cuda()
        env.step_rows_raw(a_dev.data_ptr(), prev[0].data_ptr(), prev[1].data_ptr(), nxt[0].data_ptr(), nxt[1].data_ptr())
        ora.step(act.numpy(), nthreads=8)
        torch.cuda.synchronize()
        # everything but the rows: today's f32 bands (the arena's own f32 rows are not touched by a step_rows call)
        res = compare_host_arenas(env.cfg, device_arena_to_host(env), ora.arena, what=f"bf16 rows n={n} step {t}", max_flip_frac=0.05,
                                  max_event_frac=max(2.0 / n, 1e-3), skip=("LT_F_OBS_POLICY", "LT_F_OBS_CRITIC"))
        tally.add(res)
        dones = L.arr(ora.arena, "LT_F_DONES")[:n] != 0
        n_reset += int(dones.sum())
        bad_env = np.zeros(n, bool)
        bad_env[res["flip_envs"]] = True
        bad_env[res["event_envs"]] = True
        for k, f in enumerate(("LT_F_OBS_POLICY", "LT_F_OBS_CRITIC")):
            dev = nxt[k].float().cpu().numpy()
            ref = L.arr(ora.arena, f)[:n]
            ok = np.abs(dev - ref) <= 2.0 ** -8 * np.abs(ref) + 4e-4  # bf16 rounding + the f32 observation band
            rows_bad = ~ok.all(axis=1) & ~bad_env
            assert not rows_bad.any(), (f, t, np.nonzero(rows_bad)[0][:5], float(np.abs(dev - ref)[rows_bad].max()))
            _shift_ok(prev[k].view(torch.int16).cpu().numpy(), nxt[k].view(torch.int16).cpu().numpy(), ~dones)
    env.set_row_format(torch.float32)
    print(tally.line(f"bf16 rows teacher n={n}"))
    assert n_reset > 0 and tally.events <= max(2, 5e-4 * n * 10)


def test_policy_kernel_reads_bf16_rows_exactly():
    import torch
    import torch.nn as nn
    from locotouch_amd.rl.mlp import PackedMLP

    torch.manual_seed(0)
    seq = nn.Sequential(nn.Linear(348, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, 12)).cuda()
    x16 = torch.randn(1000, 348, device="cuda:0").to(torch.bfloat16)
    a, b = PackedMLP(seq), PackedMLP(seq)
    b.set_input_format(torch.bfloat16)
    ya, yb = a(x16.float().contiguous()), b(x16)
    assert torch.equal(ya, yb)  # widening bf16 -> f32 is exact
    with pytest.raises(ValueError):
        b(x16.float())


def test_fused_rollout_with_bf16_storage_at_32768_envs():
    """Properties at config 5's size: finite outputs, every slot's history is the previous slot's shifted bit for bit (non-reset envs),
    the arena rows behind the rollout equal the last rows, and the f32 path on the same env agrees within bf16 rounding at slot 1."""
    import torch
    from locotouch_amd.env import LocoTouchVecEnv
    from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
    from tests.rl_synth import POLICY_CFG, PPO_CFG

    n, T = 32768, 6

    def make(dtype):
        env = LocoTouchVecEnv(TASK, num_envs=n, device="cuda:0", seed=3)
        torch.manual_seed(1)
        alg = PPO(ActorCritic(env.num_obs, env.num_obs, 12, **POLICY_CFG), device="cuda:0", **PPO_CFG)
        alg.init_storage(n, T, [env.num_obs], [env.num_obs], [12], obs_dtype=dtype)
        return env, alg, FusedRollout(env, alg)

    env, alg, fr = make(torch.bfloat16)
    fr.rollout(T)
    fr.rollout(T)  # slot 0 of the second rollout comes from the arena rows the first one left
    torch.cuda.synchronize()
    st = alg.storage
    assert st.observations.dtype == torch.bfloat16 and torch.isfinite(st.observations.float()).all() and torch.isfinite(st.rewards).all()
    for t in range(T - 1):
        keep = (st.dones[t, :, 0] == 0).cpu().numpy()
        _shift_ok(st.observations[t].view(torch.int16).cpu().numpy(), st.observations[t + 1].view(torch.int16).cpu().numpy(), keep)
        _shift_ok(st.privileged_observations[t].view(torch.int16).cpu().numpy(), st.privileged_observations[t + 1].view(torch.int16).cpu().numpy(), keep)
    assert torch.equal(env.obs_policy.to(torch.bfloat16), fr._tail_rows[0]) and torch.equal(env.obs_policy, fr._tail_rows[0].float())
    # against the f32 path: same seeds, same policy -> slot 0 rows equal after rounding, slot 1 within bf16 rounding of the f32 rows
    env2, alg2, fr2 = make(torch.float32)
    fr2.rollout(T)
    env3, alg3, fr3 = make(torch.bfloat16)
    fr3.rollout(T)
    torch.cuda.synchronize()
    assert torch.equal(alg3.storage.observations[0], alg2.storage.observations[0].to(torch.bfloat16))
