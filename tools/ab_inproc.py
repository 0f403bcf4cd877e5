"""Same-process alternating A/B of bench.main(): process-level A/Bs carry +-0.06 ms of run-to-run noise at the headline size,
alternation inside one process +-0.01.  Variants are environment values (read by bench.main() / the engine when they run) or Python
statements executed before each run (names: os, torch, bench, engine, L = diffalign_amd._lib; tune(skew, tail_old, stagger, prio) =
da_edge_layer_tuning).
  python tools/ab_inproc.py DA_LAST_ROWS 1,list,0 REPS [bench args]
  python tools/ab_inproc.py py "tune(-1,-1,0,0);;tune(2,-1,0,0)" REPS [bench args]"""
import os, sys, io, json, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var, reps = sys.argv[1], int(sys.argv[3])
vals = sys.argv[2].split(';;') if var == 'py' else sys.argv[2].split(',')
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-loop'] + sys.argv[4:]
import torch, bench
from diffalign_amd import engine, _lib as L


def tune(skew=-1, tail_old=-1, stagger=0, prio=0, skip_dead=1):
    L.check(L.lib().da_edge_layer_tuning(0, skew, tail_old, stagger, prio, skip_dead), 'da_edge_layer_tuning')


res = {v: [] for v in vals}
for _ in range(reps):
    for v in vals:
        if var == 'py':
            exec(v)
        else:
            os.environ[var] = v
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
        res[v].append(d['ms_per_step'])
        print(f'{v:40s} {d["ms_per_step"]:.3f} ms  edge launch {d["roofline"]["avg_launch_ms"]:.4f} ms  frac {d["roofline"]["frac"]:.4f}', flush=True)
for v in vals:
    print(f'{v:40s}: mean {sum(res[v]) / len(res[v]):.3f} ms over {len(res[v])}')
