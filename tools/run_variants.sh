#!/bin/bash
# time every experiment variant of the edge layer present in diffalign_amd/ (GPU box)
for f in diffalign_amd/libdiffalign_hip_*.so; do
  v=${f#diffalign_amd/libdiffalign_hip_}; v=${v%.so}
  echo -n "$v: "; DA_LIB=$PWD/$f timeout -k 10 100 python tools/bench_edge.py || exit 1
done
