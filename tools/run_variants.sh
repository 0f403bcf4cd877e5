#!/bin/bash
# run a micro-benchmark with every experiment variant of the library present in diffalign_amd/ (GPU box)
# usage: tools/run_variants.sh tools/bench_edge.py [args]
for f in diffalign_amd/libdiffalign_hip_*.so; do
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  echo "== $v"; DA_LIB=$PWD/$f timeout -k 10 100 python "$@" || exit 1
done
