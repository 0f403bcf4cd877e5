#!/bin/bash
# PMC passes over the edge-layer micro-benchmark (one pass per rocprofv3 run, <= 8 SQ / 4 TCC counters each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_${1:-x}
mkdir -p $OUT
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python tools/bench_edge.py 64 300 3 > $OUT/$1.log 2>&1 || echo "pass $1 failed"; }
run p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" &&
run p2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" &&
run p3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" &&
run p4 "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
python - <<'PY'
import csv, glob, collections, sys, os
out = os.environ.get('OUTDIR')
PY
for f in $(find $OUT -name "*counter_collection.csv"); do
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if 'k_edge_layer' in r['Kernel_Name']:
        agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
for k, (v, c) in sorted(agg.items()):
    print(f'{k:32s} per-launch {v / c:18.1f}  (launches {c})')
PY
done
