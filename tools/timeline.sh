#!/bin/bash
# step timeline of bench.py with the given arguments -> gpurun_out/$1 (GPU box): tools/timeline.sh NAME [bench args]
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/trace1 && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace1 -o t --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-loop "$@" > gpurun_out/trace1.log 2>&1 || exit 1
python tools/trace_step.py gpurun_out/trace1/t_kernel_trace.csv > gpurun_out/$NAME; rm -rf gpurun_out/trace1
