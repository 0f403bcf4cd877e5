"""cProfile of the host side of the benchmark's timed steps only (where the Python / ctypes time of a step goes).
python tools/host_profile.py [bench args]   e.g. --batch 100 --nodes 72"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--steps', '40', '--warmup', '5', '--no-cpu-baseline'] + sys.argv[1:]
import torch
import bench
orig_sync = torch.cuda.synchronize
segs, cur = [], [None]
def sync(*a, **k):
    if cur[0] is not None:
        cur[0].disable(); segs.append(cur[0])
    r = orig_sync(*a, **k)
    cur[0] = cProfile.Profile(); cur[0].enable()
    return r
torch.cuda.synchronize = sync
bench.main()
if cur[0] is not None: cur[0].disable()
best = max(segs, key=lambda p: pstats.Stats(p).total_tt)
s = io.StringIO()
st = pstats.Stats(best, stream=s)
st.sort_stats('tottime').print_stats(28)
print(s.getvalue()[:8000])
