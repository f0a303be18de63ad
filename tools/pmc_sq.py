"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters for lt_step_kernel<*, MODE_STEP, *> into profiles/<round>_sq_counters.json.

    python tools/pmc_sq.py <round> <key> <counter_collection.csv> [<counter_collection.csv> ...]

Per dispatch means of every counter found, plus derived: VALU instructions per wave, VALU issue utilisation
(SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES summed over SEs is not normalised here - the raw ratios below are what the guide's
profiling section uses), fraction of wave-cycles spent waiting.
"""
import csv, json, os, sys

rnd, key = sys.argv[1:3]
acc = {}
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"].replace(",0,", ", 0, ")
        if "lt_step_kernel" not in kn or ", 0, " not in kn:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locotouch_amd.build import step_kernel_source_hash  # noqa: E402  (the stamp bench.py checks before it reports these numbers)

d = {"round": rnd, "source_hash": step_kernel_source_hash(), "kernel": "lt_step_kernel (MODE_STEP)", "dispatches": {k: len(v) for k, v in acc.items()}, "mean_per_dispatch": m}
g = m.get
der = {}
if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
    der["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
if g("SQ_WAVES") and g("SQ_INSTS_SALU"):
    der["salu_insts_per_wave"] = g("SQ_INSTS_SALU") / g("SQ_WAVES")
if g("SQ_WAVES") and g("SQ_INSTS_VMEM"):
    der["vmem_insts_per_wave"] = g("SQ_INSTS_VMEM") / g("SQ_WAVES")
if g("SQ_WAVES") and g("SQ_INSTS_LDS"):
    der["lds_insts_per_wave"] = g("SQ_INSTS_LDS") / g("SQ_WAVES")
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
    der["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    der["valu_active_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
    der["mean_waves_per_busy_sq_cycle"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES")
d["derived"] = der
if g("SQ_INSTS_VALU"):
    d["valu_wave_insts_per_launch"] = g("SQ_INSTS_VALU")  # what bench.py's roofline.valu divides by the measured kernel time
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for path in (os.path.join(root, f"{rnd}_sq_counters.json"), os.path.join(root, "sq_counters.json")):  # per round + the one bench.py reads
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = d
    json.dump(data, open(path, "w"), indent=1)
print(key, json.dumps(der))
