"""Edge kernel duration by layer index inside bench.py's step (python tools/edge_by_layer.py [bench args])."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline'] + sys.argv[1:]
import torch, bench
from diffalign_amd import engine
keep = []
orig = bench.json.dumps
def dumps(o, *a, **k):
    ev = keep[0] if keep else None
    return orig(o, *a, **k)
# bench.main() sets engine.PROFILE_EVENTS = [] before the timed loop and None after: capture the list object
class Cap(list):
    pass
real_main = bench.main
import builtins
_setattr = engine.__class__.__setattr__
events = {}
def hook(name, value):
    if name == 'PROFILE_EVENTS' and isinstance(value, list):
        events['ev'] = value
import types
class M(types.ModuleType):
    def __setattr__(self, k, v):
        hook(k, v); super().__setattr__(k, v)
engine.__class__ = M
bench.main()
ev = events['ev']
torch.cuda.synchronize()
L = 9
ms = [e[0].elapsed_time(e[1]) for e in ev]
for l in range(L):
    v = ms[l::L]
    print(f'layer {l}: {sum(v) / len(v):.4f} ms  (min {min(v):.4f}, max {max(v):.4f})')
