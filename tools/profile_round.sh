# On the GPU box: kernel-trace stats and the PMC passes (each its own rocprofv3 run, --pmc never combined with a trace) of the
# bench command for the judged configurations.  Usage: bash tools/profile_round.sh <round>   -> gpurun_out/<round>prof/
set -e
RND=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${RND}prof
mkdir -p $OUT
for CFG in "4096 f32" "32768 f32" "32768 bf16"; do
  set -- $CFG; N=$1; DT=$2; TAG=${N}_${DT}
  rm -rf /tmp/kt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python3 $R/bench.py --envs $N --obs-dtype $DT --steps 240 --warmup 48 --no-cpu-baseline --update-iters 0 > $OUT/bench_under_rocprof_$TAG.json 2> $OUT/err_kt_$TAG.txt || { echo "kernel-trace FAILED $TAG"; tail -3 $OUT/err_kt_$TAG.txt; }
  cp $(ls /tmp/kt/*kernel_stats.csv | head -1) $OUT/kernel_stats_$TAG.csv 2>/dev/null || true
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    ctag=$(echo $C | tr ' ' '+' | cut -c1-40)
    rm -rf /tmp/pmc
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc -o p -- python3 $R/bench.py --envs $N --obs-dtype $DT --steps 48 --warmup 24 --no-cpu-baseline --update-iters 0 > $OUT/bench_${TAG}_${ctag}.json 2> $OUT/err_${TAG}_${ctag}.txt || { echo "FAILED $TAG $C"; tail -3 $OUT/err_${TAG}_${ctag}.txt; continue; }
    python3 - "$OUT" "$TAG" "$ctag" <<'PY'
import csv, sys, glob, os
out, tag, ctag = sys.argv[1:4]
src = glob.glob('/tmp/pmc/*counter_collection.csv')[0]
rows = [r for r in csv.DictReader(open(src)) if 'lt_step_kernel' in r['Kernel_Name'] or 'lt_mlp_kernel' in r['Kernel_Name']]
w = csv.DictWriter(open(os.path.join(out, f'cc_{tag}_{ctag}.csv'), 'w'), fieldnames=['Kernel_Name', 'Counter_Name', 'Counter_Value'])
w.writeheader()
for r in rows:
    w.writerow({k: r[k] for k in ('Kernel_Name', 'Counter_Name', 'Counter_Value')})
print(tag, ctag, len(rows))
PY
  done
done
ls $OUT | head -60
