"""ctypes mirror of include/lt_env.h and loader of the HIP library (the product path).

The `lt_cfg` structure is parsed mechanically from the header, so the Python mirror cannot drift from
the C ABI (a size check against `lt_cfg_sizeof()` guards it at load time).  There is deliberately NO
fallback: if `liblocotouch_env.so` is missing the import fails loudly - nothing on the product path
ever routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
HEADER = os.path.join(REPO, "include", "lt_env.h")
LIB_PATH = os.environ.get("LOCOTOUCH_AMD_LIB", os.path.join(_HERE, "_lib", "liblocotouch_env.so"))

_CTYPES = {"float": ctypes.c_float, "int32_t": ctypes.c_int32, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64}


def _parse_header():
    src = open(HEADER).read()
    src_nc = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src_nc = re.sub(r"//[^\n]*", "", src_nc)
    consts = {}
    for m in re.finditer(r"#define\s+(LT_\w+)\s+\(?(-?\d+)\)?\s*$", src_nc, flags=re.M):
        consts[m.group(1)] = int(m.group(2))
    # enums
    for m in re.finditer(r"enum\s+\w+\s*\{(.*?)\};", src_nc, flags=re.S):
        val = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                val = int(v, 0)
            else:
                name = item
                val += 1
            consts[name] = val
    body = re.search(r"typedef struct lt_cfg \{(.*?)\} lt_cfg;", src_nc, flags=re.S).group(1)
    fields = []
    for line in body.split(";"):
        line = line.strip()
        if not line:
            continue
        m = re.match(r"(\w+)\s+(\w+)((?:\[\w+\])*)$", line)
        if not m:
            raise RuntimeError(f"cannot parse lt_cfg member: {line!r}")
        ctype, name, dims = m.groups()
        t = _CTYPES[ctype]
        for d in reversed(re.findall(r"\[(\w+)\]", dims)):
            t = t * (int(d) if d.isdigit() else consts[d])
        fields.append((name, t))
    return consts, fields


CONSTS, _CFG_FIELDS = _parse_header()
globals().update(CONSTS)


class LtCfg(ctypes.Structure):
    _fields_ = _CFG_FIELDS

    def copy(self) -> "LtCfg":
        new = LtCfg()
        ctypes.memmove(ctypes.byref(new), ctypes.byref(self), ctypes.sizeof(self))
        # (compat/cfg_translate.py: user terms for the slow torch path - Python attributes beside the C struct)
        for attr in ("extra_reward_terms", "extra_termination_terms", "extra_observation_terms", "reward_term_params"):
            if hasattr(self, attr):
                v = getattr(self, attr)
                setattr(new, attr, dict(v) if isinstance(v, dict) else list(v))
        return new

    def to_dict(self) -> dict:
        """Plain-Python form (scalars and nested lists) for params/env.{yaml,pkl}."""
        def py(v):
            return [py(x) for x in v] if hasattr(v, "__len__") else v

        return {name: py(getattr(self, name)) for name, *_ in self._fields_ if not name.startswith("_")}


class LtView(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("dtype", ctypes.c_int32), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 3), ("stride", ctypes.c_int64 * 3)]


class LtMlpDesc(ctypes.Structure):
    _fields_ = [("num_layers", ctypes.c_int32), ("dims", ctypes.c_int32 * (CONSTS["LT_MLP_MAX_LAYERS"] + 1)), ("activation", ctypes.c_int32), ("input_format", ctypes.c_int32)]


_lib = None


def load() -> ctypes.CDLL:
    """Load liblocotouch_env.so (built in-tree by locotouch_amd.build).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is mandatory, there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.lt_abi_version.restype = ctypes.c_int
    lib.lt_cfg_sizeof.restype = ctypes.c_size_t
    lib.lt_last_error.restype = ctypes.c_char_p
    lib.lt_cfg_default.argtypes = [ctypes.c_int, ctypes.POINTER(LtCfg)]
    lib.lt_cfg_obs_dim.argtypes = [ctypes.POINTER(LtCfg)]
    lib.lt_cfg_tactile_dim.argtypes = [ctypes.POINTER(LtCfg)]
    lib.lt_cfg_preset.argtypes = [ctypes.c_char_p, ctypes.POINTER(LtCfg)]
    lib.lt_cfg_preset_id.argtypes = [ctypes.c_int]
    lib.lt_cfg_preset_id.restype = ctypes.c_char_p
    lib.lt_env_create.argtypes = [ctypes.POINTER(LtCfg), ctypes.POINTER(ctypes.c_void_p)]
    lib.lt_env_destroy.argtypes = [ctypes.c_void_p]
    lib.lt_env_state_bytes.argtypes = [ctypes.POINTER(LtCfg), ctypes.POINTER(ctypes.c_size_t)]
    lib.lt_env_bind.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lt_env_reset_all.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_step_profiled.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.lt_env_step_rows_profiled.argtypes = [ctypes.c_void_p] * 7 + [ctypes.POINTER(ctypes.c_float)]
    lib.lt_env_eval_terms.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_defer_gate.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.lt_env_gate_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_set_row_format.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.lt_env_tactile_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    _vp = ctypes.c_void_p
    lib.lt_gru_forward.argtypes = [_vp] * 5 + [ctypes.c_int] * 3 + [_vp] * 3
    lib.lt_gru_backward.argtypes = [_vp] * 6 + [ctypes.c_int] * 3 + [_vp] * 5
    lib.lt_ppo_loss.argtypes = [_vp] * 11 + [ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int] + [_vp] * 5
    lib.lt_elu_backward_bias.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp, _vp]
    lib.lt_elu_backward_bias2.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]
    lib.lt_wgrad.argtypes = [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp]
    lib.lt_split_rows.argtypes = [_vp, _vp, ctypes.c_int64, _vp]
    lib.lt_wgrad_splits.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    lib.lt_wgrad_ws_floats.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    lib.lt_wgrad_ws_floats.restype = ctypes.c_int64
    lib.lt_elu_backward_bias_ws_floats.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.lt_elu_backward_bias_ws_floats.restype = ctypes.c_int64
    lib.lt_head_wgrad.argtypes = [_vp, _vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]
    lib.lt_head_wgrad_ws_floats.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    lib.lt_head_wgrad_ws_floats.restype = ctypes.c_int64
    lib.lt_gae.argtypes = [_vp] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp]
    lib.lt_adam_clip_step.argtypes = [_vp] * 4 + [ctypes.c_int64] + [ctypes.c_float] * 6 + [ctypes.c_int64, _vp, _vp, _vp]
    lib.lt_adam_clip_step_dev.argtypes = [_vp] * 4 + [ctypes.c_int64, ctypes.c_float, _vp] + [ctypes.c_float] * 4 + [ctypes.c_int64, _vp, _vp, _vp]
    lib.lt_ppo_lr_rule.argtypes = [_vp] + [ctypes.c_float] * 4 + [_vp] * 4 + [ctypes.c_int, _vp]
    lib.lt_partial_sums.argtypes = [ctypes.c_int] + [_vp] * 8
    lib.lt_elu_backward_bias_nblk.argtypes = [ctypes.c_int64]
    lib.lt_head_wgrad_nblk.argtypes = [ctypes.c_int64]
    lib.lt_adam_clip_step_ws_floats.argtypes = [ctypes.c_int64]
    lib.lt_adam_clip_step_ws_floats.restype = ctypes.c_int64
    lib.lt_env_curriculum_apply_global.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    lib.lt_env_curriculum_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.lt_env_get_view.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(LtView)]
    lib.lt_env_set_command_ranges.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                              ctypes.c_float, ctypes.c_void_p]
    vp = ctypes.c_void_p
    lib.lt_rollout_act.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_uint64] + [vp] * 15
    lib.lt_rollout_record.argtypes = [ctypes.c_int64, ctypes.c_float] + [vp] * 9
    lib.lt_env_step_rows.argtypes = [ctypes.c_void_p] + [vp] * 6
    dp = ctypes.POINTER(LtMlpDesc)
    lib.lt_mlp_packed_floats.argtypes = [dp, ctypes.POINTER(ctypes.c_size_t)]
    lib.lt_mlp_pack.argtypes = [dp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]
    lib.lt_mlp_forward.argtypes = [dp, vp, vp, ctypes.c_int64, vp, vp]
    lib.lt_mlp_forward_pair.argtypes = [dp, vp, vp, dp, vp, vp, ctypes.c_int64, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_int, vp]
    pvp = ctypes.POINTER(vp)
    lib.lt_mlp_backward_packed_floats.argtypes = [dp, ctypes.POINTER(ctypes.c_size_t)]
    lib.lt_mlp_pack_backward.argtypes = [dp, pvp, vp, vp]
    lib.lt_mlp_pack_training.argtypes = [dp, pvp, pvp, vp, vp, dp, pvp, pvp, vp, vp, vp]
    lib.lt_mlp_backward_blocks.argtypes = [dp, dp, ctypes.c_int64]
    lib.lt_mlp_backward_blocks.restype = ctypes.c_int64
    lib.lt_mlp_backward_pair.argtypes = [dp, vp, vp, pvp, pvp, pvp, dp, vp, vp, pvp, pvp, pvp, ctypes.c_int64, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp, vp]
    lib.lt_rollout_policy.argtypes = [dp, vp, vp, ctypes.c_int64, ctypes.c_uint64, vp, ctypes.c_int64] + [vp] * 7
    lib.lt_rollout_policy_value.argtypes = [dp, vp, vp, dp, vp, vp, vp, ctypes.c_int64, ctypes.c_uint64, vp, ctypes.c_int64] + [vp] * 7
    lib.lt_env_step_rollout.argtypes = [ctypes.c_void_p] + [vp] * 6 + [ctypes.c_float, vp, vp, vp]
    lib.lt_env_kernel_name.argtypes = [ctypes.c_int]
    lib.lt_env_kernel_name.restype = ctypes.c_char_p
    if lib.lt_cfg_sizeof() != ctypes.sizeof(LtCfg):
        raise ImportError(f"lt_cfg ABI mismatch: C {lib.lt_cfg_sizeof()} vs ctypes {ctypes.sizeof(LtCfg)}")
    if lib.lt_abi_version() != CONSTS["LT_ABI_VERSION"]:
        raise ImportError("lt_env.h / liblocotouch_env.so ABI version mismatch")
    _lib = lib
    return lib


EXPORTS = ["lt_abi_version", "lt_cfg_sizeof", "lt_last_error", "lt_cfg_default", "lt_cfg_preset", "lt_cfg_num_presets", "lt_cfg_preset_id", "lt_cfg_obs_dim", "lt_cfg_tactile_dim", "lt_env_create", "lt_env_tactile_update", "lt_env_defer_gate", "lt_env_gate_update", "lt_env_check", "lt_env_set_row_format", "lt_env_step_rows_profiled", "lt_mlp_forward_pair", "lt_mlp_backward_packed_floats", "lt_mlp_pack_backward", "lt_mlp_backward_blocks", "lt_mlp_backward_pair", "lt_mlp_pack_training", "lt_env_curriculum_apply_global", "lt_gru_forward", "lt_gru_backward", "lt_ppo_loss", "lt_elu_backward_bias", "lt_elu_backward_bias_ws_floats", "lt_elu_backward_bias2", "lt_wgrad", "lt_split_rows", "lt_wgrad_splits", "lt_wgrad_ws_floats", "lt_adam_clip_step", "lt_adam_clip_step_dev", "lt_ppo_lr_rule", "lt_partial_sums", "lt_elu_backward_bias_nblk", "lt_head_wgrad_nblk", "lt_adam_clip_step_ws_floats", "lt_gae", "lt_head_wgrad", "lt_head_wgrad_ws_floats",
           "lt_env_destroy", "lt_env_state_bytes", "lt_env_bind", "lt_env_reset_all", "lt_env_step", "lt_env_step_profiled", "lt_env_eval_terms",
           "lt_env_curriculum_update", "lt_env_step_rows", "lt_env_step_rollout", "lt_env_get_view", "lt_env_set_command_ranges", "lt_rollout_act", "lt_rollout_record", "lt_mlp_packed_floats", "lt_mlp_pack", "lt_mlp_forward", "lt_rollout_policy", "lt_rollout_policy_value",
           "lt_env_kernel_name"]


def preset_ids() -> list[str]:
    lib = load()
    return [lib.lt_cfg_preset_id(i).decode() for i in range(lib.lt_cfg_num_presets())]


def preset_cfg(gym_id: str, num_envs: int | None = None, seed: int | None = None) -> LtCfg:
    """Resolved lt_cfg of a registered gym id (lt_cfg_preset)."""
    cfg = LtCfg()
    rc = load().lt_cfg_preset(gym_id.encode(), ctypes.byref(cfg))
    if rc != 0:
        raise KeyError(f"unknown task {gym_id!r}; registered: {preset_ids()}")
    if num_envs is not None:
        cfg.num_envs = int(num_envs)
    if seed is not None:
        cfg.seed = int(seed)
    return cfg


def default_cfg(task: int, num_envs: int | None = None, seed: int | None = None) -> LtCfg:
    cfg = LtCfg()
    rc = load().lt_cfg_default(task, ctypes.byref(cfg))
    if rc != 0:
        raise ValueError(f"lt_cfg_default({task}) failed: {rc}")
    if num_envs is not None:
        cfg.num_envs = num_envs
    if seed is not None:
        cfg.seed = seed
    return cfg


def check(rc: int, what: str) -> None:
    if rc != 0:
        err = load().lt_last_error()
        raise RuntimeError(f"{what} failed with code {rc}: {err.decode() if err else ''}")
