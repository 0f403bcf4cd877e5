# round-3 profile set (GPU box): headline line, kernel stats of the same command, step timeline, device-features line
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_line.json 2> gpurun_out/r3_bench.err || exit 1
BENCH_ARGS="--no-loop" bash tools/prof_bench.sh r3 > gpurun_out/r3_prof.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace1 -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loop > gpurun_out/trace1.log 2>&1 || exit 1
python tools/trace_step.py gpurun_out/trace1/t_kernel_trace.csv > gpurun_out/r3_step_timeline.txt; rm -rf gpurun_out/trace1
