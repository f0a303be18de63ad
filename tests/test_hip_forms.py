"""The step kernel's two forms against each other: four waves per 16-env tile (grids of at most one tile per CU) and one wave per
tile (larger grids; forced here with LT_STEP_HELPERS_MAX_WG=0) run the same env - same seeds, same Philox streams, the same
device functions in different waves - and must agree on everything a caller sees.  Each form is also pinned to the oracle
(test_hip_parity.py: 4096-env and 8208-env cases); this is the direct cross-check at ONE size."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(tmp_path, task, n, steps, max_wg):
    out = str(tmp_path / f"form_{max_wg}.npz")
    env = dict(os.environ)
    if max_wg is not None:
        env["LT_STEP_HELPERS_MAX_WG"] = str(max_wg)
    else:
        env.pop("LT_STEP_HELPERS_MAX_WG", None)
    r = subprocess.run([sys.executable, "-m", "tests.form_dump", task, str(n), str(steps), out], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("task,n", [("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", 1000), ("Isaac-Locomotion-LocoTouch-v1", 520)])
def test_four_wave_and_one_wave_forms_agree(tmp_path, task, n):
    steps = 6
    a = _dump(tmp_path, task, n, steps, None)   # default: n / 16 tiles <= CUs -> four-wave form
    b = _dump(tmp_path, task, n, steps, 0)      # one-wave form
    # dones and the event draws are exact (same uniforms); floats may differ by FMA contraction between the two instantiations,
    # and an env whose contact crossed a threshold in one form only is allowed to drift (counted, bounded)
    bad_envs = np.zeros(n, bool)
    for t in range(steps):
        assert np.array_equal(a[f"done{t}"], b[f"done{t}"]), f"dones differ at step {t}"
        for k in (f"obs{t}", f"critic{t}", f"rew{t}"):
            x, y = a[k].reshape(n, -1), b[k].reshape(n, -1)
            bad_envs |= (np.abs(x - y) > 2e-4 * (1.0 + np.abs(y))).any(axis=1)
    assert bad_envs.sum() <= max(2, n // 100), f"{int(bad_envs.sum())} of {n} envs differ between the forms"
    ok = ~bad_envs
    for f in ("LT_F_ROOT_POS", "LT_F_JOINT_POS", "LT_F_JOINT_VEL", "LT_F_CMD", "LT_F_EPISODE_SUMS"):
        x, y = a[f][:n][ok], b[f][:n][ok]
        # raw state, unscaled (joint velocities reach +-20 rad/s through stiff contacts; the observation carries them x 0.05)
        assert np.allclose(x, y, rtol=1e-3, atol=5e-3), (f, float(np.abs(x - y).max()))
    print(f"forms agree: {int(ok.sum())} of {n} envs within 2e-4 over {steps} steps, {int(bad_envs.sum())} drifted")
