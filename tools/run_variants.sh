#!/bin/bash
# A/B inside one GPU box: the base library first and last, every experiment variant in between.
# usage: tools/run_variants.sh tools/bench_edge.py [args]
echo "== base"; timeout -k 10 100 python "$@" || exit 1
for f in diffalign_amd/libdiffalign_hip_*.so; do
  [ -e "$f" ] || continue
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  echo "== $v"; DA_LIB=$PWD/$f timeout -k 10 100 python "$@" || exit 1
done
echo "== base"; timeout -k 10 100 python "$@"
