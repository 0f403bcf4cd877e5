#!/bin/bash
# A/B of library variants on the training step inside one GPU box: base, every diffalign_amd/libdiffalign_hip_*.so, base.
run() { n=$1; shift; timeout -k 10 300 python bench.py --workload train --steps 10 --warmup 3 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3))"; }
run base "$@" || exit 1
for f in diffalign_amd/libdiffalign_hip_*.so; do
  [ -e "$f" ] || continue
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  DA_LIB=$PWD/$f run $v "$@" || exit 1
done
run base "$@"
