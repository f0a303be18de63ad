#!/usr/bin/env python3
"""Register / LDS / scratch usage of every kernel of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.

    python tools/kernel_resources.py locotouch_amd/csrc/lt_env.hip [filter]
"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "-ffinite-math-only", "-fno-signed-zeros",
         "-freciprocal-math", "-fno-math-errno", "-mllvm", "-disable-vector-combine", "-I", os.path.join(REPO, "include")] + sys.argv[3:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?:\s+(Function Name|Name): (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if "Name:" in line:
        name = line.split("Name:")[1].split()[0]
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = name
        rows[cur] = {}
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs", "VGPR Spill"):
        if cur and f" {key}:" in line:
            rows[cur][key] = line.split(f"{key}:")[1].split()[0]
for k, v in rows.items():
    if flt in k:
        short = re.sub(r"\(anonymous namespace\)::", "", k)[:70]
        print(f"{short:72s} VGPR {v.get('VGPRs','?'):>4} AGPR {v.get('AGPRs','?'):>4} scratch {v.get('ScratchSize [bytes/lane]','?'):>5} occ {v.get('Occupancy [waves/SIMD]','?')} LDS {v.get('LDS Size [bytes/block]','?')}")
