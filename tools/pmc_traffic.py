"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/traffic.json for bench.py's roofline.traffic.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <round> [key]

Per launch of lt_step_kernel<TASK,0,*>: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 - on gfx950 FETCH_SIZE reports half of the
fetched bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section; re-checked with tools/calib_copy.hip), WRITE_SIZE is exact.
"""
import csv, json, os, sys

fetch_csv, write_csv, rnd = sys.argv[1:4]
key = sys.argv[4] if len(sys.argv) > 4 else "teacher_4096"


def mean_counter(path, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == name and "lt_step_kernel" in r["Kernel_Name"] and ", 0, " in r["Kernel_Name"].replace(",0,", ", 0, ")]
    return sum(vals) / len(vals), len(vals)


f, nf = mean_counter(fetch_csv, "FETCH_SIZE")
w, nw = mean_counter(write_csv, "WRITE_SIZE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locotouch_amd.build import step_kernel_source_hash  # noqa: E402  (the stamp bench.py checks before it reports these numbers)

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
data[key] = {"hbm_bytes_per_launch": (2 * f + w) * 1024, "round": rnd, "source_hash": step_kernel_source_hash(), "kernel": "lt_step_kernel (MODE_STEP)", "FETCH_SIZE_KB_mean": f,
             "WRITE_SIZE_KB_mean": w, "dispatches": [nf, nw],
             "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  (gfx950: FETCH_SIZE reads 1/2 of the fetched bytes - MI355X_MICROARCH.md HBM "
                        "section; re-calibrated with tools/calib_copy.hip: 1 GiB dword and float4 copies give FETCH 0.5000x, WRITE 1.0000x)",
             "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 96 --warmup 24 "
                        "--no-cpu-baseline (two separate passes)"}
json.dump(data, open(path, "w"), indent=1)
print(key, data[key]["hbm_bytes_per_launch"] / 1e6, "MB per launch", nf, nw)
