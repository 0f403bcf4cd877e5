#!/bin/bash
# time every experiment variant of the edge layer present in diffalign_amd/ (GPU box), at 1 and 2 waves per SIMD
for f in diffalign_amd/libdiffalign_hip_*.so; do
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  for t in 512 256; do
  echo -n "$v thr=$t: "; DA_EDGE_THREADS=$t DA_LIB=$PWD/$f timeout -k 10 100 python tools/bench_edge.py || exit 1
  done
done
for t in 512 256; do echo -n "base thr=$t: "; DA_EDGE_THREADS=$t timeout -k 10 100 python tools/bench_edge.py; done
