#!/bin/bash
# HBM traffic of the edge-layer launches of the benchmark step itself, per kernel instantiation (table-input first layer / layers with the
# statistics tail): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over bench.py (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-loop > $OUT/fetch.log 2>&1 || { echo "FETCH pass failed"; tail -3 $OUT/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-loop > $OUT/write.log 2>&1 || { echo "WRITE pass failed"; tail -3 $OUT/write.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for name in ('fetch', 'write'):
    for f in glob.glob(f'gpurun_out/pmc_bench/{name}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'k_edge_layer' in r['Kernel_Name']:
                inst = 'table_input' if 'true>' in r['Kernel_Name'] else 'with_tail'
                agg[(inst, r['Counter_Name'])][0] += float(r['Counter_Value']); agg[(inst, r['Counter_Name'])][1] += 1
        for (inst, k), (v, c) in sorted(agg.items()):
            print(f'{inst} {k} per launch: {v / c:.1f} (rocprofv3 unit: KB) over {c} launches')
PY
