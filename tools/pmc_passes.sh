set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02pmc
mkdir -p $OUT
for N in 4096 32768; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | tr ' ' '+' | cut -c1-40)
    rm -rf /tmp/pmc
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc -o p -- python3 $R/bench.py --envs $N --steps 48 --warmup 24 --no-cpu-baseline --update-iters 0 > $OUT/bench_${N}_${tag}.json 2> $OUT/err_${N}_${tag}.txt || { echo "FAILED $N $C"; tail -3 $OUT/err_${N}_${tag}.txt; continue; }
    python3 - "$N" "$tag" <<'PY'
import csv, sys, glob
n, tag = sys.argv[1:3]
src = glob.glob('/tmp/pmc/*counter_collection.csv')[0]
import os
out = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'r02pmc', f'cc_{n}_{tag}.csv')
rows = [r for r in csv.DictReader(open(src)) if 'lt_step_kernel' in r['Kernel_Name'] or 'lt_mlp_kernel' in r['Kernel_Name']]
w = csv.DictWriter(open(out, 'w'), fieldnames=['Kernel_Name', 'Counter_Name', 'Counter_Value'])
w.writeheader()
for r in rows:
    w.writerow({k: r[k] for k in ('Kernel_Name', 'Counter_Name', 'Counter_Value')})
print(n, tag, len(rows))
PY
  done
done
ls -la $OUT | head -40
