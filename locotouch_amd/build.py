"""In-tree build of the HIP library: hipcc --offload-arch=gfx950 -> locotouch_amd/_lib/liblocotouch_env.so.

Cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB = os.path.join(LIB_DIR, "liblocotouch_env.so")
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required)")
    return exe


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")))


def deps() -> list[str]:
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp"))
                              + glob.glob(os.path.join(REPO, "include", "*.h")))


STEP_KERNEL_SOURCES = ("lt_env.hip", "lt_physics_crba.h", "lt_post.h", "lt_device_math.h", "lt_device_prims.h")


def step_kernel_source_hash() -> str:
    """Stamp of the sources lt_step_kernel is compiled from: sha1 over their `git hash-object` ids.  The PMC summaries under
    profiles/ (traffic.json, sq_counters.json) carry the stamp of the tree they were measured on; bench.py reports their numbers
    only while the stamp still matches (a kernel change without a re-profile must not print stale counters)."""
    import hashlib

    ids = []
    for name in STEP_KERNEL_SOURCES:
        blob = open(os.path.join(CSRC, name), "rb").read()
        ids.append(hashlib.sha1(b"blob %d\0" % len(blob) + blob).hexdigest())  # == `git hash-object <file>`
    return hashlib.sha1("".join(ids).encode()).hexdigest()[:16]


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in deps())


def build_lib(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    """Compile every source to an object (device code for gfx950 only), link, and verify the offload bundle."""
    if not force and up_to_date():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    # -fno-slp-vectorize: the SLP vectoriser packs the 3-vector algebra into v_pk_* ops and pays for it with ~2400
    #   register moves per physics substep plus 1.3 KB/lane of scratch; scalar f32 code has neither (measured: -35 %).
    # -ffinite-math-only -fno-signed-zeros: lets literal-zero components of the joint offsets fold away; no NaN/Inf
    #   is ever produced on the path (divisions are guarded), and +-0 never changes a result we keep.
    # -freciprocal-math -fno-math-errno: x/y -> x*rcp(y) and bare v_sqrt (the IEEE division / sqrt expansions were 15 %
    #   of the physics phase); the ~1 ulp differences sit far inside the stated parity tolerances.
    # -mllvm -disable-vector-combine: VectorCombine widens `insertelement(poison, load float)` of the packed-pair code into
    #   `load <2 x float>` from the struct the scalar still lived in; the struct (the base position) then stayed a private
    #   array, was promoted to LDS, and the LDS slot index needs the workgroup size - a scalar load from the AQL dispatch
    #   packet in host memory: 18 k cycles (7.5 us) in front of every launch's first physics substep.
    common = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "-ffinite-math-only",
              "-fno-signed-zeros", "-freciprocal-math", "-fno-math-errno", "-mllvm", "-disable-vector-combine", "-Wall",
              "-Wno-unused-function", "-I", os.path.join(REPO, "include")] + (extra_flags or [])
    objs = []
    for s in sources():
        o = os.path.join(obj_dir, os.path.basename(s) + ".o")
        # NOTE: one translation unit per hipcc call - mixing `-x hip` and `-x c++` inputs in one command makes
        # the hipcc wrapper drop --offload-arch and emit gfx906 device code.
        cmd = [hipcc(), f"--offload-arch={ARCH}"] + common + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        objs.append(o)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    blob = open(LIB + ".tmp", "rb").read()
    if f"hipv4-amdgcn-amd-amdhsa--{ARCH}".encode() not in blob:
        raise RuntimeError(f"{LIB}: offload bundle holds no {ARCH} code object")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
