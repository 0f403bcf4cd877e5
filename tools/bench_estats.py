"""Micro-benchmark of da_edge_stats at the benchmark size: python tools/bench_estats.py [B] [n]  (chunk counts swept)"""
import sys, ctypes as C
import torch
sys.path.insert(0, '.')
from diffalign_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
E = torch.randn(B, n, n, 64, device=dev).to(torch.bfloat16)
i = lambda v: C.c_int(int(v))
for nchunk in (16, 32, 64, 128):
    st = torch.empty(B, nchunk, 4, 64, device=dev)
    for rep in range(2):
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); ev0.record()
        for _ in range(20):
            L.check(L.lib().da_edge_stats(L.ptr(E), L.ptr(st), i(B), i(n), i(64), i(nchunk), L.stream()), 'stats')
        ev1.record(); torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 20 * 1e3
    print(f'nchunk {nchunk:4d}: {us:7.1f} us  {E.numel() * 2 / us / 1e6:6.2f} TB/s')
