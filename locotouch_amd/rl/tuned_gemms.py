"""Library-GEMM algorithm selection for the PPO update's GEMM shapes (PyTorch TunableOp, results shipped in-tree).

The backward GEMMs of the update (input gradients `dz @ W`, split-K weight gradients as batched GEMMs, rl/mlp.py `backward_chain`)
are plain library GEMMs; at minibatches of 24 576 rows hipBLASLt's default heuristics pick kernels that reach 90-100 TFLOP/s f32
where other solutions of rocBLAS / hipBLASLt reach 125-150 (tools/tune_gemms.py, profiles/r04_tunableop_gfx950.txt: e.g.
`dz[24576 x 256] @ W[256 x 512]` 62 -> 50 us, the 512 x 348 weight gradient 98 -> 73 us).  `enable()` points TunableOp at the
results recorded for this GPU / ROCm / library build (`tunableop_gfx950.csv`; the file's Validator lines make TunableOp ignore
it on any other stack) with tuning itself OFF: shapes without an entry keep the default algorithm, nothing is ever tuned or
written at run time.  `LT_TUNED_GEMMS=0` disables it.
"""
from __future__ import annotations

import os

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_enabled = None


def enable() -> bool:
    global _enabled
    if _enabled is not None:
        return _enabled
    _enabled = False
    if os.environ.get("LT_TUNED_GEMMS", "1") == "0" or not torch.cuda.is_available() or not os.path.exists(RESULTS):
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):  # the user drives TunableOp themselves (e.g. tools/tune_gemms.py): hands off
        return False
    try:
        t = torch.cuda.tunable
        t.set_filename(RESULTS, insert_device_ordinal=False)
        t.tuning_enable(False)
        t.record_untuned_enable(False)
        if hasattr(t, "write_file_on_exit"):
            t.write_file_on_exit(False)
        t.enable(True)
        t.read_file(RESULTS)  # (also read lazily by the first tunable GEMM; its return value says nothing about the entries)
        _enabled = bool(t.is_enabled())
    except Exception:
        _enabled = False
    return _enabled


def disable() -> None:
    """Back to the library's default algorithm selection (tests that compare two code paths bit-tightly use it: TunableOp's
    state is per host thread - the autograd engine's worker thread does not inherit it - so the same GEMM may run a different
    algorithm in an autograd backward pass than in a direct call)."""
    global _enabled
    try:
        torch.cuda.tunable.enable(False)
    except Exception:
        pass
    _enabled = None
