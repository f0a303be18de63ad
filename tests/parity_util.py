"""Field-by-field comparison of the device arena (HIP path) with the host arena of the CPU oracle."""
from __future__ import annotations

import numpy as np

from locotouch_amd import _abi
from locotouch_amd.layout import Layout, QUAD_FIELDS

C = _abi.CONSTS

# (atol, rtol) per quad field; fp32 tolerance stated here (BASELINE.json north_star: "within a stated fp32 tolerance").
# Forces are stiff functions of position (k_n = 2e4 N/m: 1e-6 m of fp32 noise is 2e-2 N), hence the looser band there.
TOL = {
    "default": (2e-5, 2e-5),
    # positions integrate the velocities below over 20 ms: 3e-3 rad/s * 0.02 s = 6e-5
    "LT_F_JOINT_POS": (1.5e-4, 5e-5), "LT_F_ROOT_POS": (5e-5, 5e-5), "LT_F_ROOT_QUAT": (5e-5, 5e-5),
    "LT_F_OBJ_POS": (5e-5, 5e-5), "LT_F_OBJ_QUAT": (1e-4, 1e-4), "LT_F_FOOT_POS_W": (1e-4, 1e-4),
    "LT_F_ROOT_LIN_VEL_W": (2e-4, 2e-4), "LT_F_ROOT_ANG_VEL_W": (5e-4, 5e-4),
    "LT_F_OBJ_LIN_VEL_W": (2e-4, 2e-4), "LT_F_OBJ_ANG_VEL_W": (1e-3, 1e-3),
    # one env step = 4 stiff contact solves: joint velocities (|qd| up to 30 rad/s) carry ~1e-4 relative fp32 noise,
    # and joint_acc = d(qd)/dt at 200 Hz multiplies that by 200
    "LT_F_JOINT_VEL": (3e-3, 5e-4), "LT_F_JOINT_ACC": (1.0, 2e-3), "LT_F_APPLIED_TORQUE": (2e-3, 2e-4),
    "LT_F_FORCE_HIST": (5e-2, 2e-3), "LT_F_TRUNK_FORCE_HIST": (5e-2, 2e-3),
    "LT_F_FOOT_VEL_W": (2e-3, 1e-3),
    "LT_F_EPISODE_SUMS": (2e-4, 2e-4), "LT_F_LAST_EPISODE_SUMS": (2e-4, 2e-4), "LT_F_REWARD_TERMS": (2e-3, 5e-4),
    "LT_F_CURRICULUM": (2e-4, 2e-4),
    # tactile plate samples: contact point (x, y) in the trunk frame [position band], normal force [force band]
    "LT_F_PLATE_SAMPLES": (5e-5, 5e-5),
    # lanes 2, 3: the command term's error_vel_xy / error_vel_yaw - functions of the root velocities (bands above)
    "LT_F_EVENT_TIMERS": (5e-4, 5e-4), "LT_F_LAST_CMD_METRICS": (5e-4, 5e-4),
}
# (field, lane) pairs that hold foot_air_time_variance: a function of the thresholded contact timers, so flip-tolerant like them
AIR_VARIANCE_LANES = {"LT_F_TRUNK_FORCE_HIST": 3, "LT_F_LAST_CMD_METRICS": 2}
PLATE_FORCE_TOL = (5e-2, 2e-3)
# binary taxel map: a taxel whose force sits within fp32 noise of its threshold may flip; more than this many differing
# taxels in one env is a failure, not a flip
MAX_TAXEL_FLIPS_PER_ENV = 4
# per-term overrides inside LT_F_REWARD_TERMS (unweighted terms): the L2 norms of joint acc / vel / torque inherit the
# tolerance of their inputs
TERM_TOL = {C["LT_R_JOINT_ACCELERATION"]: (2.0, 2e-3), C["LT_R_JOINT_VELOCITY"]: (5e-3, 5e-4), C["LT_R_JOINT_TORQUE"]: (5e-3, 5e-4),
            C["LT_R_FOOT_SLIP"]: (4e-3, 1e-3)}
# fields whose values hinge on a thresholded contact force (|F| > 1 N): a borderline flip changes the timer by a whole dt.
FLIP_TOLERANT = {"LT_F_FOOT_CUR_AIR", "LT_F_FOOT_CUR_CONTACT", "LT_F_FOOT_LAST_AIR", "LT_F_FOOT_LAST_CONTACT", "LT_F_OBJ_TIMERS",
                 "LT_F_GAIT_LAST_AIR", "LT_F_GAIT_LAST_CONTACT", "LT_F_GAIT_VALID_LAST_AIR", "LT_F_GAIT_FLAGS"}


def device_arena_to_host(env) -> np.ndarray:
    return env._arena_aligned.detach().cpu().numpy().copy()


def compare_host_arenas(cfg, dev: np.ndarray, ref: np.ndarray, what: str = "", max_flip_frac: float = 0.0, obs_tol=(4e-4, 4e-4),
                        skip=(), max_event_frac: float = 0.0):
    """`max_flip_frac`: envs allowed to differ in thresholded contact booleans / timers (force within fp32 noise of 1 N).
    `max_event_frac`: envs allowed to diverge beyond the fp32 band in continuous fields this step - a discontinuous event
    (joint-limit clamp zeroing a velocity, contact (de)activation) taken on one side and not on the other because the
    deciding quantity sat within rounding of its threshold.  Callers bound the total over a run as well."""
    n = int(cfg.num_envs)
    obs_dim = (45 if cfg.task == C["LT_TASK_LOCOMOTION"] else 58) * int(cfg.obs_history)
    wide = int(cfg.tactile_format) in (C["LT_TACTILE_PROCESSED"], C["LT_TACTILE_ORIGINAL"])
    L = Layout(n, obs_dim, int(cfg.tactile_enabled), C["LT_TACTILE_WIDE_DIM"] if wide else C["LT_TACTILE_DIM"])
    assert dev.shape == ref.shape == (L.total_bytes,), (dev.shape, ref.shape, L.total_bytes)
    report, failures, flip_envs, event_envs, soft_envs = [], [], set(), set(), {}
    for name in QUAD_FIELDS:
        if name in skip or (name == "LT_F_PLATE_SAMPLES" and not L.tactile):
            continue
        a, b = L.vec(dev, name), L.vec(ref, name)
        if name == "LT_F_GAIT_FLAGS":
            bad = (a.view(np.int32) != b.view(np.int32)).any(axis=1)
            err = float(bad.mean())
        else:
            atol, rtol = TOL.get(name, TOL["default"])
            if name == "LT_F_REWARD_TERMS":
                atol = np.full(a.shape[1], atol, np.float32)
                rtol = np.full(a.shape[1], rtol, np.float32)
                for col, (ta, tr) in TERM_TOL.items():
                    atol[col], rtol[col] = ta, tr
            if name == "LT_F_PLATE_SAMPLES":  # columns 8..11 = normal force of the four samples
                atol = np.array([atol] * 8 + [PLATE_FORCE_TOL[0]] * 4, np.float32)
                rtol = np.array([rtol] * 8 + [PLATE_FORCE_TOL[1]] * 4, np.float32)
            with np.errstate(invalid="ignore"):
                badm = ~(np.abs(a - b) <= atol + rtol * np.abs(b))
            badm |= ~np.isfinite(a)
            if name in AIR_VARIANCE_LANES:
                col = AIR_VARIANCE_LANES[name]
                off = ~(np.abs(a[:, col] - b[:, col]) <= 2e-5 + 2e-5 * np.abs(b[:, col]))
                flip_envs |= set(np.nonzero(off)[0].tolist())
                badm[:, col] = False
            bad = badm.any(axis=1)
            err = float(np.nanmax(np.abs(a - b))) if a.size else 0.0
        report.append((name, err, int(bad.sum())))
        if bad.any():
            if name in FLIP_TOLERANT:
                flip_envs |= set(np.nonzero(bad)[0].tolist())
            else:
                failures.append((name, err, np.nonzero(bad)[0][:5].tolist()))
                event_envs |= set(np.nonzero(bad)[0].tolist())
    hard = []  # failures that no allowance covers
    for name in L.plain:
        if name in skip or name.startswith("_"):
            continue
        a, b = L.arr(dev, name), L.arr(ref, name)
        if name in ("LT_F_OBS_TACTILE", "LT_F_OBS_TACTILE_ORIGINAL", "LT_F_OBS_TACTILE_PROCESSED"):
            bit = {"LT_F_OBS_TACTILE": 4, "LT_F_OBS_TACTILE_ORIGINAL": 1, "LT_F_OBS_TACTILE_PROCESSED": 2}[name]
            if not L.tactile or not ((int(cfg.tactile_aux_groups) | 4) & bit):
                continue
            a, b = a[:n], b[:n]
            nt = C["LT_TACTILE_ROWS"] * C["LT_TACTILE_COLS"]
            assert np.isin(a[:, :nt], (0.0, 1.0)).all(), "the contact channel must be binary"
            per_env = (a[:, :nt] != b[:, :nt]).sum(axis=1)  # flipped taxels of the contact channel
            bad = per_env > 0
            err = float(per_env.max()) if n else 0.0
            if (per_env > MAX_TAXEL_FLIPS_PER_ENV).any():
                hard.append((name, err, np.nonzero(per_env > MAX_TAXEL_FLIPS_PER_ENV)[0][:5].tolist()))
            else:
                flip_envs |= set(np.nonzero(bad)[0].tolist())
            # the force channels (normalised / min-max / discretised) of envs whose contact map agrees: fp32 band (a flipped
            # taxel moves the env's min-max range and with it every other taxel's value)
            same = ~bad
            dv = np.abs(a[same, nt:] - b[same, nt:])
            if dv.size and not (dv <= 2e-4 + 2e-4 * np.abs(b[same, nt:])).all():
                # a discretisation level is 1 / total_levels: a rounding-boundary flip of single taxels is a flip, more is a failure
                lvl = (dv > 2e-4 + 2e-4 * np.abs(b[same, nt:])).sum(axis=1)
                if (lvl > MAX_TAXEL_FLIPS_PER_ENV).any():
                    hard.append((name + " (force channels)", float(dv.max()), np.nonzero(same)[0][lvl > MAX_TAXEL_FLIPS_PER_ENV][:5].tolist()))
                else:
                    flip_envs |= set(np.nonzero(same)[0][lvl > 0].tolist())
            report.append((name, err, int(bad.sum())))
            continue
        if name in ("LT_F_OBS_POLICY", "LT_F_OBS_CRITIC", "LT_F_REWARD"):
            a, b = a[:n], b[:n]
            atol, rtol = obs_tol
            badm = ~(np.abs(a - b) <= atol + rtol * np.abs(b)) | ~np.isfinite(a)
            bad = badm.reshape(n, -1).any(axis=1)
            err = float(np.abs(a - b).max())
            if bad.any():
                soft_envs[name] = (err, set(np.nonzero(bad)[0].tolist()))
        elif name == "LT_F_GATE_RING":  # population sums of the last passes: float sums formed in a different order on each side
            bad = np.array([not np.allclose(a, b, rtol=2e-5, atol=2e-4)])
            err = float(np.abs(a - b).max())
            if bad.any():
                hard.append((name, err, []))
        elif name == "LT_F_CMD_PARAMS":
            bad = np.array([not np.allclose(a, b, atol=1e-6)])
            err = float(np.abs(a - b).max())
            if bad.any():
                hard.append((name, err, []))
        elif name == "LT_F_COUNTERS":
            bad = np.array([a[0] != b[0] or a[3] != b[3]])
            err = float(abs(int(a[0]) - int(b[0])))
            if bad.any():
                hard.append((name, err, []))
        else:  # integer outputs: bit-exact (reset / episode indexing)
            a, b = a[:n], b[:n]
            bad = a != b
            err = float(bad.mean())
            if np.any(bad):
                if name in ("LT_F_TERM_BITS", "LT_F_DONES", "LT_F_TERMINATED"):
                    # a termination term decided by a thresholded contact force (|F| history vs 1 N) may flip with it
                    flip_envs |= set(np.nonzero(bad)[0].tolist())
                else:
                    hard.append((name, err, np.nonzero(bad)[0][:5].tolist()))
        report.append((name, err, int(np.sum(bad))))
    # Observation rows / reward may differ beyond the fp32 band ONLY in envs that, in this very step, flipped a thresholded
    # contact quantity (FLIP_TOLERANT field, termination bit, dones) or took a discontinuous event (hard-field divergence,
    # which includes a thresholded reward term in LT_F_REWARD_TERMS).  Anything else is a plain failure.
    explained = flip_envs | event_envs
    for name, (err, envs) in soft_envs.items():
        rest = sorted(envs - explained)
        if rest:
            hard.append((name + " (no contact flip / event in these envs)", err, rest[:5]))
    n_flip, n_event = len(flip_envs - event_envs), len(event_envs)
    over = []
    if n_event / max(1, n) > max_event_frac:
        over.append(f"{n_event}/{n} envs diverge in continuous fields (allowed fraction {max_event_frac:.4g})")
    if n_flip / max(1, n) > max_flip_frac:
        over.append(f"{n_flip}/{n} envs differ in thresholded fields (allowed fraction {max_flip_frac:.4g})")
    if hard or over:
        lines = [f"parity {what}: {len(hard)} hard failures; " + "; ".join(over)]
        lines += [f"  HARD {nm}: max|err|={e:.3e} envs={ids}" for nm, e, ids in hard]
        lines += [f"  EVENT {nm}: max|err|={e:.3e} envs={ids}" for nm, e, ids in failures]
        lines += [f"  {nm:28s} max|err|={e:.3e} bad_envs={nb}" for nm, e, nb in report if nb]
        raise AssertionError("\n".join(lines))
    return dict(report=report, flip_envs=sorted(flip_envs - event_envs), event_envs=sorted(event_envs),
                forgiven_obs_envs=sorted(set().union(*[e for _, e in soft_envs.values()])) if soft_envs else [])


class Tally:
    """Running count of what the allowances forgave over a multi-step run (printed by the tests, so a regression shows)."""

    def __init__(self, n: int):
        self.n, self.steps, self.flips, self.events, self.obs = n, 0, 0, 0, 0

    def add(self, res: dict) -> None:
        self.steps += 1
        self.flips += len(res["flip_envs"])
        self.events += len(res["event_envs"])
        self.obs += len(res["forgiven_obs_envs"])

    def line(self, what: str) -> str:
        tot = max(1, self.n * self.steps)
        return (f"[parity] {what}: {self.steps} steps x {self.n} envs; forgiven thresholded-contact flips {self.flips} "
                f"({self.flips / tot:.2e}), discontinuous events {self.events} ({self.events / tot:.2e}), "
                f"obs/reward rows outside the band in those envs {self.obs}")


def compare_arenas(env, ora, what: str = "", **kw):
    return compare_host_arenas(env.cfg, device_arena_to_host(env), ora.arena, what=what, **kw)
