"""include/lt_go1_model.h is GENERATED from the reference's URDFs by tools/compile_robot_model.py: regenerating it from the
read-only checkout must reproduce the committed header byte for byte (CPU; skipped where the checkout does not exist)."""
import os
import subprocess
import sys

import pytest

REF_URDF = "/root/reference/locotouch/utils/urdf_processor/go1/urdf/locotouch_without_tactile.urdf"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(REF_URDF), reason="reference checkout not present")
def test_robot_model_header_regenerates_identically():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "compile_robot_model.py")], capture_output=True, text=True, check=True).stdout
    committed = open(os.path.join(REPO, "include", "lt_go1_model.h")).read()
    assert out == committed or out.rstrip("\n") == committed.rstrip("\n")
    # the taxel grid the tactile kernel uses is the URDF's (17 x 13 boxes of 18.3 x 17.5 mm on a 14.3 x 12.8 mm pitch)
    for line in ("#define LT_TAXEL_ROWS 17", "#define LT_TAXEL_COLS 13", "#define LT_TAXEL_DX 0.0143f", "#define LT_TAXEL_HALF_Y 0.00875f"):
        assert line in committed
