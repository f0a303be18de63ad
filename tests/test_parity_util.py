"""The parity helper itself (CPU): it must FAIL on an observation / reward mismatch that no thresholded-contact flip or
discontinuous event explains, and forgive only what its allowances name."""
import numpy as np
import pytest

from locotouch_amd import _abi
from locotouch_amd.layout import Layout
from tests import oracle_lib as O
from tests.parity_util import Tally, compare_host_arenas

C = _abi.CONSTS


def _pair(n=32, steps=3):
    cfg = _abi.default_cfg(C["LT_TASK_TRANSPORT_TEACHER"], num_envs=n, seed=3)
    cfg.debug_terms = 1
    ora = O.OracleEnv(cfg)
    ora.reset_all()
    rng = np.random.default_rng(0)
    for _ in range(steps):
        ora.step((0.3 * rng.standard_normal((n, 12))).astype(np.float32))
    return cfg, ora.arena.copy(), Layout(n, 348)


def test_identical_arenas_pass_with_zero_allowance():
    cfg, a, _ = _pair()
    res = compare_host_arenas(cfg, a.copy(), a)
    assert res["flip_envs"] == [] and res["event_envs"] == [] and res["forgiven_obs_envs"] == []


@pytest.mark.parametrize("field", ["LT_F_OBS_POLICY", "LT_F_OBS_CRITIC", "LT_F_REWARD"])
def test_unexplained_obs_or_reward_mismatch_is_a_hard_failure(field):
    cfg, a, L = _pair()
    b = a.copy()
    v = L.arr(b, field)
    if field == "LT_F_REWARD":
        v[5] += 0.01
    else:
        v.reshape(-1, 348)[5, 17] += 0.01
    with pytest.raises(AssertionError, match="no contact flip"):
        compare_host_arenas(cfg, b, a, max_flip_frac=0.5, max_event_frac=0.5)  # generous allowances must not hide it


def test_obs_mismatch_is_forgiven_only_in_an_env_that_flipped():
    cfg, a, L = _pair()
    b = a.copy()
    L.arr(b, "LT_F_OBS_POLICY").reshape(-1, 348)[5, 17] += 0.01
    t = L.vec(b, "LT_F_FOOT_CUR_AIR")
    t[5, 0] += 0.005  # a whole sensor period: the contact boolean flipped in env 5
    L.set_vec(b, "LT_F_FOOT_CUR_AIR", t)
    with pytest.raises(AssertionError):  # zero allowance
        compare_host_arenas(cfg, b, a)
    res = compare_host_arenas(cfg, b, a, max_flip_frac=1.0 / 32)
    assert res["flip_envs"] == [5] and res["forgiven_obs_envs"] == [5]
    # the same obs error in a DIFFERENT env than the flip is not forgiven
    c = b.copy()
    L.arr(c, "LT_F_OBS_POLICY").reshape(-1, 348)[6, 3] += 0.01
    with pytest.raises(AssertionError, match="no contact flip"):
        compare_host_arenas(cfg, c, a, max_flip_frac=0.5)


def test_event_allowance_is_bounded_and_counted():
    cfg, a, L = _pair()
    b = a.copy()
    q = L.vec(b, "LT_F_JOINT_VEL")
    q[2, 1] += 0.5
    L.set_vec(b, "LT_F_JOINT_VEL", q)
    with pytest.raises(AssertionError):
        compare_host_arenas(cfg, b, a, max_flip_frac=0.5)  # events have their own allowance
    res = compare_host_arenas(cfg, b, a, max_event_frac=1.0 / 32)
    assert res["event_envs"] == [2]
    tally = Tally(32)
    tally.add(res)
    assert "discontinuous events 1" in tally.line("x")


def test_integer_outputs_stay_bit_exact():
    cfg, a, L = _pair()
    b = a.copy()
    L.arr(b, "LT_F_EP_LEN")[4] += 1
    with pytest.raises(AssertionError, match="LT_F_EP_LEN"):
        compare_host_arenas(cfg, b, a, max_flip_frac=0.5, max_event_frac=0.5)
