#!/usr/bin/env python3
"""Static instruction profile of one kernel from a `-gline-tables-only -S` listing: per basic block (label to label) the VALU /
SALU / LDS / VMEM counts and the source lines (innermost inlined location) that own the VALU instructions.

    hipcc ... -gline-tables-only --cuda-device-only -S csrc/lt_env.hip -o /tmp/x.s
    python tools/asm_profile.py /tmp/x.s <mangled-kernel-substring> [--blocks] [--top N]
"""
import collections, re, sys

path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
show_blocks = "--blocks" in sys.argv
files = {}
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", l)
    if m:
        files[int(m.group(1))] = m.group(3).split("/")[-1]
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"\s+md5", l)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
    if start is None and re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l):
        start = i
assert start is not None, "kernel not found"
cur = ("?", 0)
blocks = []  # (label, counts, per-line)
blk = ["entry", collections.Counter(), collections.Counter()]
tot = collections.Counter()
by_line = collections.Counter()
def kind(op):
    if op.startswith("v_"):
        if "mfma" in op: return "MFMA"
        return "VALU"
    if op.startswith("s_"):
        if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "SYNC"
        if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_memtime"): return "SMEM"
        if "branch" in op: return "BR"
        return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
    return "OTHER"
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.Lfunc_end", l):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append(blk)
        blk = [m.group(1), collections.Counter(), collections.Counter()]
        continue
    m = re.match(r"^\t([a-z_0-9]+)\b", l)
    if not m or l.startswith("\t."):
        continue
    k = kind(m.group(1))
    blk[1][k] += 1
    tot[k] += 1
    if k == "VALU":
        blk[2][cur] += 1
        by_line[cur] += 1
        tot["op:" + re.sub(r"_e(32|64)$", "", m.group(1))] += 1
blocks.append(blk)
print("total", {k: v for k, v in tot.items() if not k.startswith("op:")})
ops = sorted(((v, k[3:]) for k, v in tot.items() if k.startswith("op:")), reverse=True)
print("VALU ops:", ", ".join(f"{k} {v}" for v, k in ops[:30]))
if show_blocks:
    for lab, cnt, per in blocks:
        if cnt["VALU"] >= 40:
            print(f"{lab:14s} VALU {cnt['VALU']:5d} SALU {cnt['SALU']:4d} LDS {cnt['LDS']:4d} VMEM {cnt['VMEM']:4d}  top:",
                  ", ".join(f"{f}:{ln}x{n}" for (f, ln), n in per.most_common(6)))
print("--- VALU by source line (innermost) ---")
for (f, ln), n in by_line.most_common(top):
    print(f"{n:6d}  {f}:{ln}")
