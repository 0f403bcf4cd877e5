"""Micro-benchmark of da_edge_layer alone (for rocprofv3 --pmc passes).  python tools/bench_edge.py [B] [n] [iters] [tail]
(tail: da_edge_layer_stats, the Etoy statistics computed by the kernel's tail)"""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from diffalign_amd import engine, _lib as L
from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
from diffalign_amd import synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D = torch.device('cuda:0')
hd = {'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128}
torch.manual_seed(0)
m = GraphTransformerWithYAtomMapPosEmb(1, {'X': 37, 'E': 7, 'y': 13}, {'X': 256, 'E': 128, 'y': 128}, hd, {'X': 29, 'E': 7, 'y': 1},
                                       nn.ReLU(), nn.ReLU(), p_to_r_skip_connection=True, p_to_r_init=1.0, alignment_type='correct').eval().to(D)
pm = m._pack(D)
g = torch.Generator(device=D).manual_seed(1)
X = torch.randn(B, n, 256, device=D, generator=g)
E = torch.randn(B, n, n, 64, device=D, generator=g).to(torch.bfloat16)
y = torch.randn(B, 64, device=D, generator=g)
nm = torch.ones(B, n, dtype=torch.uint8, device=D)
st = None if (len(sys.argv) > 4 and sys.argv[4] == 'tail') else engine.edge_stats(E, B, n, 64)
for _ in range(2):
    engine.layer_forward(pm, pm.layers[0], X, E, y, nm, st, B, n)
torch.cuda.synchronize()
engine.PROFILE_EVENTS = []
for _ in range(iters):
    engine.layer_forward(pm, pm.layers[0], X, E, y, nm, st, B, n)
torch.cuda.synchronize()
ms = sum(e[0].elapsed_time(e[1]) for e in engine.PROFILE_EVENTS) / iters
pairs = B * n * n
print(f'edge_layer B={B} n={n}: {ms:.3f} ms/launch  {pairs*131072/ms/1e9:.1f} TFLOP/s  {pairs*256/ms/1e6:.1f} GB/s algorithmic')
