#!/bin/bash
# HBM traffic of the edge-layer kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (GPU box)
# usage: tools/pmc_traffic.sh [tail]   (tail: the kernel with its statistics tail, da_edge_layer_stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic$1
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python tools/bench_edge.py 64 300 3 $1 > $OUT/fetch.log 2>&1 || { echo "FETCH pass failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python tools/bench_edge.py 64 300 3 $1 > $OUT/write.log 2>&1 || { echo "WRITE pass failed"; exit 1; }
python - $1 <<'PY'
import csv, glob, collections, sys
for name in ('fetch', 'write'):
    for f in glob.glob(f'gpurun_out/pmc_traffic{sys.argv[1] if len(sys.argv) > 1 else ""}/{name}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'k_edge_layer' in r['Kernel_Name']:
                agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
        for k, (v, c) in agg.items():
            print(f'{k} per k_edge_layer launch: {v / c:.1f} (rocprofv3 unit: KB)  over {c} launches')
PY
