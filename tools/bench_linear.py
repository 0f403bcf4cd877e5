"""Micro-benchmark of da_linear: python tools/bench_linear.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffalign_amd import engine, pack, _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
D = torch.device('cuda:0')
g = torch.Generator(device=D).manual_seed(1)
for (K, N, relu) in [(256, 256, False), (256, 768, False), (256, 256, True), (37, 256, True), (256, 29, False)]:
    A = torch.randn(M, K, device=D, generator=g)
    W = torch.randn(N, K, device=D, generator=g) * 0.05
    b = torch.randn(N, device=D, generator=g)
    Wp = pack.bf16_weight(W.cpu()).to(D)
    out = torch.empty(M, N, device=D)
    for _ in range(3):
        engine._linear(A, K, Wp, b, None, out, N, M, N, K, relu)
    ref = A @ W.t() + b
    if relu: ref = ref.relu()
    err = ((out - ref).norm() / ref.norm()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        engine._linear(A, K, Wp, b, None, out, N, M, N, K, relu)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'M={M} K={K} N={N} relu={relu}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s  {(M*K*4+M*N*4)/us/1e3:.0f} GB/s  relerr {err:.2e}')
