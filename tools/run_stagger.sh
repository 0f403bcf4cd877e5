#!/bin/bash
for st in 0 4 8 12 16 24; do for pr in 0 1; do
  echo -n "stagger=$st prio=$pr: "; DA_EDGE_STAGGER=$st DA_EDGE_PRIO=$pr timeout -k 10 100 python tools/bench_edge.py || exit 1
done; done
