timeout -k 10 300 python bench.py > gpurun_out/bench_line_d.json 2> gpurun_out/bench_d.err; cut -c150-200 gpurun_out/bench_line_d.json
BENCH_ARGS="" bash tools/prof_bench.sh d > gpurun_out/prof_d.txt 2>&1; head -1 gpurun_out/prof_d.txt | cut -c150-200
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload uspto_like > gpurun_out/bench_line_d_uspto_like.json; cut -c150-200 gpurun_out/bench_line_d_uspto_like.json
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extra-features device > gpurun_out/bench_line_d_device_features.json; cut -c150-200 gpurun_out/bench_line_d_device_features.json
DA_STATS_TAIL=0 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_line_d_bare_edge_kernel.json
python -c "import json; d=json.load(open('gpurun_out/bench_line_d_bare_edge_kernel.json')); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
timeout -k 10 200 python tools/bench_sampling.py > gpurun_out/bench_sampling_d.txt 2>&1; tail -1 gpurun_out/bench_sampling_d.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace1 -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/trace1.log 2>&1
python tools/trace_step.py gpurun_out/trace1/t_kernel_trace.csv > gpurun_out/step_timeline_d.txt; rm -rf gpurun_out/trace1
