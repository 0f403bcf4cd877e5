#!/bin/bash
# Issue-level PMC counters of the edge-layer kernel (two passes of <= 8 SQ counters each) over the micro-benchmark (GPU box):
# how busy the matrix pipe is, how much of that overlaps vector work, where the waves wait.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_issue
mkdir -p $OUT
A="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $OUT/a -- python3 tools/bench_edge.py 64 300 3 > $OUT/a.log 2>&1 || { echo "pass A failed"; tail -3 $OUT/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $OUT/b -- python3 tools/bench_edge.py 64 300 3 > $OUT/b.log 2>&1 || { echo "pass B failed"; tail -3 $OUT/b.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for name in ('a', 'b'):
    for f in glob.glob(f'gpurun_out/pmc_issue/{name}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'k_edge_layer' in r['Kernel_Name']:
                agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
        for k, (v, c) in sorted(agg.items()):
            res[k] = v / c
            print(f'{k} per launch: {v / c:.4g} over {c} launches')
json.dump(res, open('gpurun_out/pmc_issue/summary.json', 'w'), indent=1)
PY
