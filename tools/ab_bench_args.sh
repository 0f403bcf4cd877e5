#!/bin/bash
# A/B of library variants on the whole step inside one GPU box with extra bench.py arguments: base, every variant, base.
# usage: tools/ab_bench_args.sh --batch 100 --nodes 72
run() { n=$1; shift; timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loop "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3))"; }
run base "$@" || exit 1
for f in diffalign_amd/libdiffalign_hip_*.so; do
  [ -e "$f" ] || continue
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  DA_LIB=$PWD/$f run $v "$@" || exit 1
done
run base "$@"
