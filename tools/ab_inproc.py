"""Same-process alternating A/B of an environment switch that bench.main() reads when it runs: process-level A/Bs carry +-0.06 ms of
run-to-run noise at the headline size, alternation inside one process +-0.02.
python tools/ab_inproc.py VAR v1,v2[,v3] REPS [bench args]"""
import os, sys, io, json, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var, vals, reps = sys.argv[1], sys.argv[2].split(','), int(sys.argv[3])
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-loop'] + sys.argv[4:]
import bench
res = {v: [] for v in vals}
for _ in range(reps):
    for v in vals:
        os.environ[var] = v
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
        res[v].append(d['ms_per_step'])
        print(f'{var}={v} {d["ms_per_step"]:.3f} ms  edge launch {d["roofline"]["avg_launch_ms"]:.4f} ms  frac {d["roofline"]["frac"]:.4f}', flush=True)
for v in vals:
    print(f'{var}={v}: mean {sum(res[v]) / len(res[v]):.3f} ms over {len(res[v])}')
