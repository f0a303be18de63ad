"""ctypes loader for the CPU oracle (oracle/_build/liblt_oracle.so) - test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from locotouch_amd import _abi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(REPO, "oracle", "_build", "liblt_oracle.so")

f32 = ctypes.c_float
i32 = ctypes.c_int32


class TermIn(ctypes.Structure):
    _fields_ = [("root_pos", f32 * 3), ("root_quat", f32 * 4), ("root_lin", f32 * 3), ("root_ang", f32 * 3),
                ("q", f32 * 3 * 4), ("qd", f32 * 3 * 4), ("qdd", f32 * 3 * 4), ("tau", f32 * 3 * 4),
                ("act_raw", f32 * 3 * 4), ("act_prev", f32 * 3 * 4),
                ("fhist", f32 * 4 * 4 * 3), ("trunk_fhist", f32 * 3),
                ("foot_pos", f32 * 3 * 4), ("foot_vel", f32 * 3 * 4),
                ("obj_pos", f32 * 3), ("obj_quat", f32 * 4), ("obj_lin", f32 * 3), ("obj_ang", f32 * 3),
                ("obj_timers", f32 * 4), ("cmd", f32 * 3), ("terminated", i32)]


class GaitIO(ctypes.Structure):
    _fields_ = [("cur_air", f32 * 4), ("cur_con", f32 * 4), ("sensor_last_air", f32 * 4), ("cmd", f32 * 3),
                ("lin_err", f32), ("ang_err", f32), ("obj_xy_yaw", f32 * 2), ("any_nonzero_cmd", i32),
                ("last_step_air", f32 * 4), ("last_step_con", f32 * 4), ("valid_last_air", f32 * 4),
                ("swinging_in_zero_cmd", i32 * 4), ("valid_prev_contact", i32 * 4), ("last_cmd", f32 * 3),
                ("step_from_change", f32)]


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(REPO, "oracle", "lt_oracle.c")):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(ORACLE_SO)
    P = ctypes.POINTER
    lib.lt_oracle_obs_dim.argtypes = [P(_abi.LtCfg)]
    lib.lt_oracle_state_bytes.argtypes = [P(_abi.LtCfg)]
    lib.lt_oracle_state_bytes.restype = ctypes.c_int64
    lib.lt_oracle_reset_all.argtypes = [P(_abi.LtCfg), ctypes.c_void_p]
    lib.lt_oracle_step.argtypes = [P(_abi.LtCfg), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.lt_oracle_eval_terms.argtypes = [P(_abi.LtCfg), ctypes.c_void_p]
    lib.lt_oracle_process_action.argtypes = [P(_abi.LtCfg), P(f32), P(f32), P(f32), P(f32)]
    lib.lt_oracle_gait.argtypes = [P(_abi.LtCfg), P(GaitIO), f32]
    lib.lt_oracle_gait.restype = f32
    lib.lt_oracle_rewards.argtypes = [P(_abi.LtCfg), P(TermIn), P(GaitIO), f32, P(f32)]
    lib.lt_oracle_terminations.argtypes = [P(_abi.LtCfg), P(TermIn), ctypes.c_int64, ctypes.c_int64]
    lib.lt_oracle_object_state_obs.argtypes = [P(_abi.LtCfg), P(TermIn), P(f32), P(f32)]
    lib.lt_oracle_command_update.argtypes = [ctypes.c_int64, ctypes.c_int, P(f32), ctypes.c_int, P(f32)]
    lib.lt_oracle_cmd_params_init.argtypes = [P(_abi.LtCfg), P(f32)]
    lib.lt_oracle_curriculum.argtypes = [P(_abi.LtCfg), P(f32), ctypes.c_int64, P(f32), P(f32)]
    lib.lt_oracle_command_resample_u.argtypes = [P(_abi.LtCfg), P(f32), P(f32), P(f32), f32, f32, P(f32), P(f32), P(f32), P(f32)]
    lib.lt_oracle_material_u.argtypes = [P(f32), P(f32), P(f32), P(f32), P(f32)]
    lib.lt_oracle_taxel_forces.argtypes = [P(f32), P(f32), P(f32), P(f32)]
    lib.lt_oracle_tactile_signals_u.argtypes = [P(_abi.LtCfg)] + [P(f32)] * 5
    lib.lt_oracle_tactile_channels_u.argtypes = [P(_abi.LtCfg), ctypes.c_int] + [P(f32)] * 13
    lib.lt_oracle_tactile_format_u.argtypes = [P(_abi.LtCfg), ctypes.c_int, P(f32), P(P(f32)), P(f32)]
    lib.lt_oracle_reset_object_u.argtypes = [P(_abi.LtCfg)] + [P(f32)] * 4 + [f32] + [P(f32)] * 5
    lib.lt_oracle_curriculum_update.argtypes = [P(_abi.LtCfg), ctypes.c_void_p, P(f32)]
    lib.lt_oracle_gate_on_sums.argtypes = [P(_abi.LtCfg), P(f32), P(f32), f32, ctypes.c_int, ctypes.c_int, P(i32)]
    lib.lt_oracle_curriculum_apply_global.argtypes = [P(_abi.LtCfg), ctypes.c_void_p, P(f32), ctypes.c_int, ctypes.c_int64]
    lib.lt_oracle_obs_push.argtypes = [P(i32), ctypes.c_int, ctypes.c_int, P(f32), ctypes.c_int, P(f32)]
    _lib = lib
    return lib


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(f32))


class OracleEnv:
    """Whole-env oracle on a host arena with the device layout (include/lt_layout.h)."""

    def __init__(self, cfg: _abi.LtCfg):
        self.cfg = cfg.copy()
        self.lib = load()
        self.nbytes = self.lib.lt_oracle_state_bytes(ctypes.byref(self.cfg))
        self.arena = np.zeros(self.nbytes, dtype=np.uint8)

    @property
    def ptr(self):
        return self.arena.ctypes.data_as(ctypes.c_void_p)

    def reset_all(self):
        self.lib.lt_oracle_reset_all(ctypes.byref(self.cfg), self.ptr)

    def step(self, actions: np.ndarray, nthreads: int = 1):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        self.lib.lt_oracle_step(ctypes.byref(self.cfg), self.ptr, a.ctypes.data_as(ctypes.c_void_p), nthreads)

    def curriculum_update(self, records: np.ndarray):
        r = np.ascontiguousarray(records, dtype=np.float32)
        self.lib.lt_oracle_curriculum_update(ctypes.byref(self.cfg), self.ptr, fptr(r))

    def curriculum_apply_global(self, ring_sums: np.ndarray, nsteps: int, n_total: int):
        r = np.ascontiguousarray(ring_sums, dtype=np.float32)
        self.lib.lt_oracle_curriculum_apply_global(ctypes.byref(self.cfg), self.ptr, fptr(r), int(nsteps), int(n_total))

    def eval_terms(self):
        self.lib.lt_oracle_eval_terms(ctypes.byref(self.cfg), self.ptr)
