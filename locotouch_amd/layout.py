"""Python mirror of include/lt_layout.h (state-arena layout) for host arenas and tests.

The product path takes its views from `lt_env_get_view` (C); tests/test_abi.py checks that this mirror and
the C layout agree field by field.
"""
from __future__ import annotations

import numpy as np

from . import _abi

C = _abi.CONSTS

_QUADS3 = {"LT_F_JOINT_POS", "LT_F_JOINT_VEL", "LT_F_JOINT_ACC", "LT_F_APPLIED_TORQUE", "LT_F_ACT_RAW", "LT_F_ACT_PREV_RAW",
           "LT_F_ACT_PREV_PREV_RAW", "LT_F_FOOT_POS_W", "LT_F_FOOT_VEL_W", "LT_F_CURRICULUM", "LT_F_PLATE_SAMPLES"}
_QUADS7 = {"LT_F_EPISODE_SUMS", "LT_F_LAST_EPISODE_SUMS", "LT_F_REWARD_TERMS"}
QUAD_FIELDS = [k for k, v in sorted(C.items(), key=lambda kv: kv[1]) if k.startswith("LT_F_") and v < C["LT_NUM_QUAD_FIELDS"]]


def field_quads(name: str) -> int:
    if name in _QUADS3:
        return 3
    if name in _QUADS7:
        return 7
    if name == "LT_F_FORCE_HIST":
        return 12
    return 1


def _align(x: int) -> int:
    return (x + 255) & ~255


class Layout:
    def __init__(self, num_envs: int, obs_dim: int, tactile: int = 0, tactile_dim: int | None = None):
        """`tactile`: cfg.tactile_enabled; `tactile_dim`: width of the `tactile` group (442, or 884 for the 4-channel formats)."""
        self.n = num_envs
        self.tactile = int(bool(tactile))
        self.tactile_dim = int(tactile_dim or C["LT_TACTILE_DIM"])
        self.npad = (num_envs + 15) // 16 * 16
        self.obs_dim = obs_dim
        off = 0
        self.quad_off = {}
        for name in QUAD_FIELDS:
            self.quad_off[name] = off
            quads = 0 if (name == "LT_F_PLATE_SAMPLES" and not self.tactile) else field_quads(name)
            off = _align(off + quads * self.npad * 16)
        self.plain = {}
        for name, nbytes, dtype, shape in [
            ("LT_F_EP_LEN", self.npad * 8, np.int64, (self.npad,)),
            ("LT_F_OBS_POLICY", self.npad * obs_dim * 4, np.float32, (self.npad, obs_dim)),
            ("LT_F_OBS_CRITIC", self.npad * obs_dim * 4, np.float32, (self.npad, obs_dim)),
            ("LT_F_REWARD", self.npad * 4, np.float32, (self.npad,)),
            ("LT_F_DONES", self.npad * 8, np.int64, (self.npad,)),
            ("LT_F_TERMINATED", self.npad, np.uint8, (self.npad,)),
            ("LT_F_TIME_OUT", self.npad, np.uint8, (self.npad,)),
            ("LT_F_TERM_BITS", self.npad * 4, np.int32, (self.npad,)),
            ("LT_F_CMD_PARAMS", C["LT_CMD_PARAMS_LEN"] * 4, np.float32, (C["LT_CMD_PARAMS_LEN"],)),
            ("LT_F_COUNTERS", 4 * 8, np.int64, (4,)),
            ("LT_F_GATE_RING", C["LT_GATE_RING"] * 8 * 4, np.float32, (C["LT_GATE_RING"], 8)),
            ("_PARTIALS", 2 * (self.npad // 16) * 8 * 4, np.float32, (2 * (self.npad // 16), 8)),  # per-tile curriculum partials, two sets (step parity)
            # three blocks of [npad][884] capacity: tactile | original_tactile | processed_tactile (include/lt_layout.h)
            ("LT_F_OBS_TACTILE", 3 * self.npad * C["LT_TACTILE_WIDE_DIM"] * 4 * self.tactile, np.float32, (self.npad * self.tactile, self.tactile_dim)),
            ("LT_F_OBJ_SIZES", self.npad * 2 * 4, np.float32, (self.npad, 2)),
            ("_DEV_ARGS", 4096, np.uint8, (4096,)),  # device copy of (lt_cfg, lt_layout), include/lt_layout.h
        ]:
            self.plain[name] = (off, dtype, shape)
            if name == "LT_F_OBS_TACTILE":
                blk, wide = self.npad * C["LT_TACTILE_WIDE_DIM"] * 4, (self.npad * self.tactile, C["LT_TACTILE_WIDE_DIM"])
                self.plain["LT_F_OBS_TACTILE_ORIGINAL"] = (off + blk, dtype, wide)
                self.plain["LT_F_OBS_TACTILE_PROCESSED"] = (off + 2 * blk, dtype, wide)
            off = _align(off + nbytes)
        self.total_bytes = off

    def quad(self, arena: np.ndarray, name: str) -> np.ndarray:
        """View [Q][npad][4] (float32) of a quad field inside a uint8 host arena."""
        q = field_quads(name)
        off = self.quad_off[name]
        return arena[off:off + q * self.npad * 16].view(np.float32).reshape(q, self.npad, 4)

    def vec(self, arena: np.ndarray, name: str) -> np.ndarray:
        """Copy of a quad field as [n][Q*4] with component c = q*4 + lane."""
        v = self.quad(arena, name)
        return np.ascontiguousarray(v.transpose(1, 0, 2).reshape(self.npad, -1)[: self.n])

    def set_vec(self, arena: np.ndarray, name: str, values: np.ndarray) -> None:
        v = self.quad(arena, name)
        q = v.shape[0]
        vals = np.zeros((self.npad, q * 4), dtype=np.float32)
        vals[: values.shape[0], : values.shape[1]] = values
        v[:] = vals.reshape(self.npad, q, 4).transpose(1, 0, 2)

    def arr(self, arena: np.ndarray, name: str) -> np.ndarray:
        off, dtype, shape = self.plain[name]
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return arena[off:off + nbytes].view(dtype).reshape(shape)
