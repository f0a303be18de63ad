"""The step kernel's physics source (csrc/lt_physics_crba.h, packed-pair formulation) runs on the CPU through tools/host_twin -
four lock-stepped threads stand for a quad's lanes - and is compared with the frozen scalar formulation on random states that
exercise every contact sphere.  Tolerances sit at the noise floor of the scalar formulation itself (its own outputs on inputs
perturbed in the last bit: 1.2e-4 after one substep, 4e-4 after four; TWIN_SELF=1 measures it)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(REPO, "tools", "host_twin")


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or ""):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def twin(tmp_path_factory):
    cc = _clang()
    if cc is None:
        pytest.skip("no clang++ (ext_vector_type) for the host twin")
    exe = str(tmp_path_factory.mktemp("twin") / "twin")
    subprocess.run([cc, "-std=c++20", "-O1", "-I", TWIN, "-I", os.path.join(REPO, "include"), '-DLT_PRIMS_H="twin_prims.h"',
                    os.path.join(TWIN, "twin.cpp"), os.path.join(REPO, "locotouch_amd", "csrc", "lt_cfg.cpp"), "-lpthread", "-o", exe],
                   check=True)
    return exe


@pytest.mark.parametrize("nstates,nsub,tol", [(400, 1, 5e-4), (200, 4, 2e-3)])
def test_packed_physics_matches_scalar_formulation(twin, nstates, nsub, tol):
    out = subprocess.run([twin, str(nstates), str(nsub), str(tol)], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    for line in out.stdout.splitlines():  # every sphere type must have been active somewhere, or the comparison proves little
        if "active contacts" in line and nsub == 1:
            counts = [int(t) for t in line.split("active contacts")[1].replace(",", " ").split() if t.isdigit()]
            assert all(c > 0 for c in counts[:4]), line
