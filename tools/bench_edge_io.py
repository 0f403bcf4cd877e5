"""Micro-benchmark of da_edge_in / da_edge_out at the benchmark size: python tools/bench_edge_io.py [B] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from diffalign_amd import engine, _lib as L
from diffalign_amd.engine import _i
from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
D = torch.device('cuda:0')
hd = {'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128}
torch.manual_seed(0)
m = GraphTransformerWithYAtomMapPosEmb(1, {'X': 37, 'E': 7, 'y': 13}, {'X': 256, 'E': 128, 'y': 128}, hd, {'X': 29, 'E': 7, 'y': 1},
                                       nn.ReLU(), nn.ReLU(), p_to_r_skip_connection=True, p_to_r_init=1.0, alignment_type='correct').eval().to(D)
pm = m._pack(D)
g = torch.Generator(device=D).manual_seed(1)
cls = torch.randint(0, 7, (B, n, n), device=D, generator=g)
cls = torch.triu(cls, 1); cls = cls + cls.transpose(1, 2)
E = torch.nn.functional.one_hot(cls, 7).float().contiguous()
nm = torch.ones(B, n, dtype=torch.uint8, device=D)
Eh = torch.empty(B, n, n, 64, dtype=torch.bfloat16, device=D)
Eo = torch.empty(B, n, n, 7, device=D)
src = torch.full((B, n), -1, dtype=torch.int32, device=D)
s1, s2 = pm.skip
lib = L.lib()
def run_in():
    L.check(lib.da_edge_in(L.ptr(E), _i(7), L.ptr(nm), L.ptr(pm.in_E), L.ptr(Eh), _i(B), _i(n), _i(64), _i(128), L.stream()), 'in')
def run_out():
    L.check(lib.da_edge_out(L.ptr(Eh), L.ptr(E), _i(7), L.ptr(pm.out_E), L.ptr(src), L.ptr(s1), L.ptr(s2), L.ptr(Eo), _i(B), _i(n),
                            _i(64), _i(128), _i(7), L.stream()), 'out')
for name, fn in (('edge_in', run_in), ('edge_out', run_out)):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name} B={B} n={n}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us')
