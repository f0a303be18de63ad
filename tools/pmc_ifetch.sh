# instruction-fetch / wait counters of the step kernel (separate --pmc passes; run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02ifetch
mkdir -p $OUT
for N in 4096 32768; do
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
    tag=$(echo $C | tr ' ' '+' | cut -c1-40)
    rm -rf /tmp/pmc
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc -o p -- python3 $R/bench.py --envs $N --steps 48 --warmup 24 --no-cpu-baseline --update-iters 0 > $OUT/bench_${N}_${tag}.json 2> $OUT/err_${N}_${tag}.txt || { echo "FAILED $N $C"; tail -3 $OUT/err_${N}_${tag}.txt; continue; }
    python3 - "$N" <<'PY'
import csv, sys, glob, collections
n = sys.argv[1]
src = glob.glob('/tmp/pmc/*counter_collection.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    if 'lt_step_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(n, k, 'mean_per_dispatch %.1f' % (sum(v) / len(v)), 'dispatches', len(v), flush=True)
PY
  done
done
