#!/bin/bash
# rocprofv3 kernel-trace of bench.py on the GPU box -> gpurun_out/prof_$1/ (kernel_stats csv)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-x}
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | cut -c1-260
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
