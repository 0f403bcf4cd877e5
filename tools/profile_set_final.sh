#!/bin/bash
# The whole committed profile set of a round in one GPU-box call -> gpurun_out/final/ (copy into profiles/rNN/ afterwards):
# PMC traffic of the edge kernel first (bench.py reads its json), then the driver's headline command, the profiled line + kernel stats,
# the other sizes, the step timelines and the training line + kernel stats.  Every step has its own time limit; the chain stops at the
# first failure.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
bash tools/pmc_traffic_bench.sh > $O/pmc_bench.txt 2>&1 || { tail -5 $O/pmc_bench.txt; exit 1; }
python tools/traffic_json.py $O/pmc_bench.txt > /dev/null || exit 1
cp profiles/r03/edge_layer_traffic.json $O/
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err || exit 1
BENCH_ARGS="--no-loop" bash tools/prof_bench.sh final > $O/prof.txt 2>&1 || exit 1
grep '^{' gpurun_out/prof_final/bench.log > $O/bench_line_profiled.json
cp $(find gpurun_out/prof_final -name '*kernel_stats.csv' | sort | tail -1) $O/bench_kernel_stats.csv
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loop --batch 100 --nodes 72 > $O/bench_line_n72.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loop --batch 64 --nodes 140 > $O/bench_line_n140.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --workload uspto_like > $O/bench_line_uspto_like.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --extra-features device > $O/bench_line_device_features.json 2>> $O/bench.err || exit 1
bash tools/timeline.sh final/step_timeline.txt || exit 1
bash tools/timeline.sh final/step_timeline_device_features.txt --extra-features device || exit 1
bash tools/timeline.sh final/step_timeline_n72_graph.txt --batch 100 --nodes 72 || exit 1
timeout -k 10 300 python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2>> $O/bench.err || exit 1
bash tools/prof_train.sh final > $O/prof_train.txt 2>&1 || exit 1
cp $(find gpurun_out/prof_train_final -name '*kernel_stats.csv' | sort | tail -1) $O/train_kernel_stats.csv
for f in bench_line bench_line_profiled bench_line_n72 bench_line_n140 bench_line_uspto_like bench_line_device_features train_line; do
  python -c "import json,sys; d=json.load(open('$O/$f.json')); r=d.get('roofline'); r=r if isinstance(r,dict) else {}; print('$f', round(d['ms_per_step'],3), r.get('frac'), r.get('traffic'))"
done
