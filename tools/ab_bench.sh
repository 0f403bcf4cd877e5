#!/bin/bash
# A/B of library variants on the whole step inside one GPU box: base, every diffalign_amd/libdiffalign_hip_*.so, base.
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['value']), 'edge_us', round(d['roofline'].get('launch_us', 0) or 0, 1))"; }
run base || exit 1
for f in diffalign_amd/libdiffalign_hip_*.so; do
  [ -e "$f" ] || continue
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  DA_LIB=$PWD/$f run $v || exit 1
done
run base
