import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline'] + sys.argv[1:]
import bench
# monkeypatch: wrap torch.cuda.synchronize to report the host time of the timed loop
orig_sync = torch.cuda.synchronize
marks = []
def sync(*a, **k):
    marks.append(time.perf_counter())
    r = orig_sync(*a, **k)
    marks.append(time.perf_counter())
    return r
torch.cuda.synchronize = sync
bench.main()
# pairs: (before sync, after sync); the timed region lies between the 'after' of one sync and the 'before' of the next
for i in range(1, len(marks) - 1, 2):
    print('host enqueue %.2f ms, then waited %.2f ms for the GPU' % ((marks[i + 1] - marks[i]) * 1e3, (marks[i + 2] - marks[i + 1]) * 1e3))
from diffalign_amd.noise_schedule import DiffusionTransition
print('step_tables:', DiffusionTransition.STEP_STATS)
