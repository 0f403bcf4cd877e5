#!/bin/bash
# rocprofv3 kernel-trace of the training step (bench.py --workload train) on the GPU box -> gpurun_out/prof_train_$1/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_${1:-x}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 2 $BENCH_ARGS > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | cut -c1-260
python $GRAFT_REPO_ROOT/tools/kstats.py $OUT 7 40
