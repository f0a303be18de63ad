"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/lt_env.h declares,
the ctypes mirror matches, and the Python layout mirror agrees with lt_env_get_view.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from locotouch_amd import _abi
from locotouch_amd.layout import Layout, QUAD_FIELDS, field_quads

C = _abi.CONSTS


def test_library_exports_every_declared_symbol():
    lib = _abi.load()
    src = re.sub(r"/\*.*?\*/", "", open(_abi.HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(lt_\w+)\s*\(", src)) - {"lt_align256"}
    declared = {d for d in declared if not d.startswith("lt_field") and not d.startswith("lt_layout") and d != "lt_quad"}
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_cfg_struct_mirror_and_defaults():
    lib = _abi.load()
    assert lib.lt_cfg_sizeof() == ctypes.sizeof(_abi.LtCfg)
    loco = _abi.default_cfg(C["LT_TASK_LOCOMOTION"])
    teach = _abi.default_cfg(C["LT_TASK_TRANSPORT_TEACHER"])
    assert lib.lt_cfg_obs_dim(ctypes.byref(loco)) == 270 and lib.lt_cfg_obs_dim(ctypes.byref(teach)) == 348
    assert loco.num_envs == 4096 and loco.decimation == 4 and abs(loco.sim_dt - 0.005) < 1e-9 and loco.max_episode_length == 1000
    # 17 active rewards for locomotion, 23 for the teacher
    assert sum(1 for i in range(C["LT_NUM_REWARD_TERMS"]) if loco.reward_weight[i] != 0) == 17
    assert sum(1 for i in range(C["LT_NUM_REWARD_TERMS"]) if teach.reward_weight[i] != 0) == 23
    assert [teach.term_enabled[i] for i in range(7)] == [1, 1, 1, 0, 1, 1, 1]
    assert [loco.term_enabled[i] for i in range(7)] == [1, 1, 1, 1, 1, 0, 0]
    assert teach.cmd_multi_sampling == 1 and teach.cur_enabled == 1 and loco.cur_enabled == 0
    assert lib.lt_cfg_default(7, ctypes.byref(loco)) == C["LT_EINVAL"]


@pytest.mark.parametrize("task,n,tactile,fmt,aux", [(0, 4096, 0, 0, 0), (1, 4096, 0, 0, 0), (1, 100, 0, 0, 0), (1, 17, 0, 0, 0), (1, 405, 1, 0, 0),
                                                    (1, 20, 1, 0, 3), (1, 33, 1, 4, 2), (1, 16, 1, 2, 1)])
def test_layout_mirror_matches_c(task, n, tactile, fmt, aux):
    lib = _abi.load()
    cfg = _abi.default_cfg(task, num_envs=n)
    cfg.tactile_enabled = tactile
    cfg.tactile_format, cfg.tactile_aux_groups = fmt, aux
    tdim = lib.lt_cfg_tactile_dim(ctypes.byref(cfg))
    assert tdim == (0 if not tactile else 884 if fmt in (4, 5) else 442)
    h = ctypes.c_void_p()
    assert lib.lt_env_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    nbytes = ctypes.c_size_t()
    assert lib.lt_env_state_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)) == 0
    L = Layout(n, lib.lt_cfg_obs_dim(ctypes.byref(cfg)), tactile, tdim or None)
    assert L.total_bytes == nbytes.value
    v = _abi.LtView()
    for name in QUAD_FIELDS:
        if name == "LT_F_PLATE_SAMPLES" and not tactile:  # the tactile fields exist only with cfg.tactile_enabled
            assert lib.lt_env_get_view(h, C[name], ctypes.byref(v)) == C["LT_EINVAL"]
            continue
        assert lib.lt_env_get_view(h, C[name], ctypes.byref(v)) == 0
        assert (v.ptr or 0) == L.quad_off[name], name
        assert list(v.shape) == [n, field_quads(name), 4] and list(v.stride) == [4, L.npad * 4, 1]
    for name, (off, dtype, shape) in L.plain.items():
        if name.startswith("_"):
            continue  # internal regions (device args block) have no public view
        bit = {"LT_F_OBS_TACTILE": 4, "LT_F_OBS_TACTILE_ORIGINAL": 1, "LT_F_OBS_TACTILE_PROCESSED": 2}.get(name)
        if bit and (not tactile or not ((aux | 4) & bit)):  # the tactile groups exist only when enabled
            assert lib.lt_env_get_view(h, C[name], ctypes.byref(v)) == C["LT_EINVAL"]
            continue
        assert lib.lt_env_get_view(h, C[name], ctypes.byref(v)) == 0
        assert (v.ptr or 0) == off, name
        if bit:
            assert list(v.shape)[:2] == [n, shape[1]] and list(v.stride)[:2] == [shape[1], 1], name
    assert lib.lt_env_get_view(h, 63, ctypes.byref(v)) == C["LT_EINVAL"]
    rc = lib.lt_env_get_view(h, C["LT_F_OBS_OBJECT_STATE"], ctypes.byref(v))  # a strided window of the policy rows
    if task == 1:
        assert rc == 0 and (v.ptr or 0) == L.plain["LT_F_OBS_POLICY"][0] + 270 * 4 and list(v.shape)[:2] == [n, 78] and list(v.stride)[:2] == [348, 1]
    else:
        assert rc == C["LT_EINVAL"]
    # error behaviour: stepping / resetting without a bound arena fails loudly, never silently
    assert lib.lt_env_reset_all(h, None) == C["LT_EFAULT"]
    dummy = (ctypes.c_float * 12)()
    assert lib.lt_env_step(h, ctypes.cast(dummy, ctypes.c_void_p), None) == C["LT_EFAULT"]
    assert b"not bound" in lib.lt_last_error()
    assert lib.lt_env_bind(h, ctypes.c_void_p(256), 16) == C["LT_EFAULT"]
    assert lib.lt_env_destroy(h) == 0


def test_invalid_cfg_rejected():
    lib = _abi.load()
    cfg = _abi.default_cfg(1)
    cfg.num_envs = 0
    h = ctypes.c_void_p()
    assert lib.lt_env_create(ctypes.byref(cfg), ctypes.byref(h)) == C["LT_EINVAL"]
    cfg.num_envs = 8
    cfg.obs_history = 3
    assert lib.lt_env_create(ctypes.byref(cfg), ctypes.byref(h)) == C["LT_EINVAL"]


def test_product_path_has_no_oracle_dependency():
    """The shipped package must never import / link the oracle (it is the checker, not the product)."""
    import glob
    import os

    root = os.path.dirname(_abi.__file__)
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True) + glob.glob(os.path.join(root, "csrc", "*")):
        text = open(path, errors="ignore").read()
        # comments may cite the oracle as the executable spec; code may not include, import, link or dlopen it
        assert not re.search(r'#include\s+"[^"]*oracle', text), path
        assert not re.search(r"oracle_lib|liblt_oracle|from\s+oracle|import\s+oracle|lt_oracle_\w+\s*\(", text), path
    _ = np


def test_update_and_recurrence_entry_points_validate_their_arguments_before_touching_the_gpu():
    """Error behaviour of the PPO-update / GRU entry points (include/lt_env.h): a bad argument is LT_EINVAL with a message, decided
    on the host - no launch, so this runs without a GPU."""
    lib = _abi.load()
    vp = ctypes.c_void_p
    one = vp(16)  # any non-null address: argument checks come before the first dereference / launch
    null = vp(None)
    bad = C["LT_EINVAL"]
    # lt_ppo_loss: null operand, zero rows, more than 16 actions
    args = [one] * 10 + [null]
    assert lib.lt_ppo_loss(*([null] + args[1:]), 128, 12, 0.2, 1.0, 0.01, 1, one, one, one, one, null) == bad
    assert lib.lt_ppo_loss(*args, 0, 12, 0.2, 1.0, 0.01, 1, one, one, one, one, null) == bad
    assert lib.lt_ppo_loss(*args, 128, 17, 0.2, 1.0, 0.01, 1, one, one, one, one, null) == bad
    assert b"lt_ppo_loss" in lib.lt_last_error()
    # lt_elu_backward_bias: N not a multiple of 4 / beyond 1024
    assert lib.lt_elu_backward_bias(one, one, 64, 130, 1.0, one, one, one, null) == bad
    assert lib.lt_elu_backward_bias(one, one, 64, 2048, 1.0, one, one, one, null) == bad
    assert lib.lt_elu_backward_bias_ws_floats(96, 512) == 2 * 512 and lib.lt_elu_backward_bias_ws_floats(97, 512) == 3 * 512
    # lt_head_wgrad: more than 16 outputs, k not a multiple of 4, k beyond 1024
    assert lib.lt_head_wgrad(one, one, 0, 64, 17, 128, one, one, one, null) == bad
    assert lib.lt_head_wgrad(one, one, 0, 64, 12, 130, one, one, one, null) == bad
    assert lib.lt_head_wgrad(one, one, 0, 64, 16, 2048, one, one, one, null) == bad and b"lt_head_wgrad" in lib.lt_last_error()
    assert lib.lt_head_wgrad_ws_floats(96, 12, 128) == 12 * 128 + 16
    # lt_adam_clip_step: empty buffer, step 0
    assert lib.lt_adam_clip_step(one, one, one, one, 0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, one, null, null) == bad
    assert lib.lt_adam_clip_step(one, one, one, one, 10, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, one, null, null) == bad
    assert lib.lt_adam_clip_step_ws_floats(2048) == 1 and lib.lt_adam_clip_step_ws_floats(2049) == 2
    # GRU: hidden size must be a multiple of 64
    assert lib.lt_gru_forward(one, one, one, one, one, 4, 8, 96, one, one, null) == bad
    assert lib.lt_gru_backward(one, null, one, one, one, one, 4, 8, 96, one, one, one, one, null) == bad
    assert b"multiple of 64" in lib.lt_last_error()


def test_population_pass_placement_api():
    """lt_env_defer_gate / lt_env_gate_update (include/lt_env.h): modes 0..2, nothing to do while no pass is outstanding, no launch
    without a bound arena."""
    lib = _abi.load()
    cfg = _abi.default_cfg(1, num_envs=64)
    h = ctypes.c_void_p()
    assert lib.lt_env_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    for mode in (0, 1, 2, 0):
        assert lib.lt_env_defer_gate(h, mode) == 0
    assert lib.lt_env_defer_gate(h, 4) == C["LT_EINVAL"] and lib.lt_env_defer_gate(h, -1) == C["LT_EINVAL"]  # (3 = mode 2 + the lost-announcement test hook)
    assert lib.lt_env_defer_gate(None, 0) == C["LT_EINVAL"]
    assert lib.lt_env_gate_update(h, None) == C["LT_EFAULT"] and b"not bound" in lib.lt_last_error()
    assert lib.lt_env_destroy(h) == 0


def test_train_script_dumps_env_and_agent_params(tmp_path):
    """params/{env,agent}.{yaml,pkl} (reference locotouch/scripts/train.py:150-153): the resolved lt_cfg and the agent cfg as dicts."""
    import pickle

    import yaml

    from locotouch_amd import _abi as A
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.scripts.train import dump_params

    task = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"
    cfg = A.preset_cfg(task, num_envs=128)
    dump_params(str(tmp_path), dict(cfg.to_dict(), gym_id=task), train_cfg(task))
    assert sorted(os.listdir(tmp_path / "params")) == ["agent.pkl", "agent.yaml", "env.pkl", "env.yaml"]
    env = yaml.safe_load(open(tmp_path / "params" / "env.yaml"))
    assert env["num_envs"] == 128 and env["gym_id"] == task and len(env["reward_weight"]) >= 25
    assert pickle.load(open(tmp_path / "params" / "env.pkl", "rb")) == env
    assert pickle.load(open(tmp_path / "params" / "agent.pkl", "rb")) == yaml.safe_load(open(tmp_path / "params" / "agent.yaml"))
