#!/bin/bash
# EXPERIMENT: do two half-batch pipelines on one GPU overlap each other's gaps?  Two bench processes at B=32 side by side against one at B=64.
cd $GRAFT_REPO_ROOT
j() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'graph-steps/s')"; }
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loop 2>/dev/null | j "one process B=64" || exit 1
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loop --batch 32 2>/dev/null | j "one process B=32" || exit 1
timeout -k 10 300 python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-loop --batch 32 > gpurun_out/two_a.json 2>/dev/null &
PA=$!
timeout -k 10 300 python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-loop --batch 32 > gpurun_out/two_b.json 2>/dev/null &
PB=$!
wait $PA || exit 1
wait $PB || exit 1
j "two processes, A (B=32)" < gpurun_out/two_a.json; j "two processes, B (B=32)" < gpurun_out/two_b.json
