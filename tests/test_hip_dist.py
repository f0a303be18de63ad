"""Multi-rank readiness of the path the GPU actually runs (rl/ppo.py `_fused_update` + `_optim_step`: packed forward, lt_ppo_loss,
flat gradient bucket -> all-reduce (mean) -> lt_adam_clip_step; KL all-reduced before the learning-rate decision; advantage
moments all-reduced): two ranks on ONE card over gloo (RCCL refuses two ranks per device; 8-GPU runs are the driver's).
tests/test_dist_gloo.py covers the op-chain update on the CPU; this is the fused one."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, out, minibatches):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LT_DIST_BACKEND="gloo", LT_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", LT_TEST_MINIBATCHES=str(minibatches))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_dist_child.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0


def test_two_rank_fused_update_replicas_are_bit_equal_and_match_one_process(tmp_path):
    out = str(tmp_path)
    for mb in (1, 2):
        _run(2, out, mb)
        _run(1, out, mb)
    for mb in (1, 2):
        r0, r1, one = (np.load(os.path.join(out, f)) for f in (f"rank0of2_mb{mb}.npz", f"rank1of2_mb{mb}.npz", f"rank0of1_mb{mb}.npz"))
        np.testing.assert_array_equal(r0["params"], r1["params"])  # replicas stay bit-identical through every optimizer step
        assert float(r0["lr"]) == float(r1["lr"])
        assert np.all(np.isfinite(r0["losses"])) and np.all(np.isfinite(one["losses"]))
        d = np.abs(r0["params"] - one["params"])
        if mb == 1:
            # one minibatch per epoch: the mean of the two shards' gradients is the single process's full-batch gradient, the KL and
            # the advantage moments are all-reduced -> the same update up to the rounding of differently ordered sums
            assert float(r0["lr"]) == float(one["lr"])
            # (Adam divides by sqrt(v): a parameter whose gradient sits at the rounding level can move by a visible fraction of one
            # step either way - a handful out of 687 513 - so: all within one learning-rate step, all but 1e-5 of them tight)
            tight = d <= 3e-5 + 3e-4 * np.abs(one["params"])
            assert d.max() < 1e-3 and (~tight).mean() < 1e-5, (d.max(), int((~tight).sum()))
        else:
            # several minibatches: each rank permutes its own shard, the partitions differ -> agreement to the size of an Adam step
            assert d.max() < 8e-3 and d.mean() < 1.5e-3, (d.max(), d.mean())
