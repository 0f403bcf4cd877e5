"""GPU micro-benchmark of the training-path kernels at the training size (B=16, n=140, align_absorbing widths).
   python tools/bench_backward.py [B n]     (DA_ATTN_BWD=<rows><slices> picks the attention-backward variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffalign_amd import _lib as L
from diffalign_amd.autograd_ops import AttnCore, AddLayerNorm, PoolStats, WideLinear

B, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 140)
dev = 'cuda:0'
dx, de = 256, 64


def timed(name, fn, bytes_, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f'{name:34s} {ms * 1e3:8.1f} us   {bytes_ / ms / 1e6:7.1f} GB/s algorithmic ({bytes_ / 1e6:.0f} MB)', flush=True)


g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
nm = (torch.arange(n, device=dev)[None] < torch.randint(n // 3, n + 1, (B, 1), device=dev, generator=g)).to(torch.uint8)
Q, K, V, E1, E2, ye1, ye2 = r(B, n, dx), r(B, n, dx), r(B, n, dx), r(B, n, n, dx), r(B, n, n, dx), r(B, dx), r(B, dx)
big = B * n * n * dx * 4
Z = torch.empty_like(E1); wV = torch.empty_like(Q); sm = torch.empty_like(Q); sl = torch.empty_like(Q)
fwd = lambda: L.check(L.lib().da_attn_core_fwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm), L.ptr(ye1), L.ptr(ye2), 0.177, L.ptr(Z),
                                               L.ptr(wV), L.ptr(sm), L.ptr(sl), B, n, dx, L.stream()), 'fwd')
timed('attn_core_fwd', fwd, 3 * big)
dZ, dwV = r(B, n, n, dx), r(B, n, dx)
wsA = torch.empty(int(L.lib().da_attn_core_bwd_workspace(B, n, dx)), dtype=torch.uint8, device=dev)
dQ, dK, dV, dE1, dE2, d1, d2 = (torch.empty_like(t) for t in (Q, Q, Q, E1, E1, ye1, ye1))
bwd = lambda: L.check(L.lib().da_attn_core_bwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm), L.ptr(ye2), 0.177, L.ptr(wV), L.ptr(sm),
                                               L.ptr(sl), L.ptr(dZ), L.ptr(dwV), L.ptr(dQ), L.ptr(dK), L.ptr(dV), L.ptr(dE1), L.ptr(dE2), L.ptr(d1),
                                               L.ptr(d2), L.ptr(wsA), wsA.numel(), B, n, dx, L.stream()), 'bwd')
timed(f'attn_core_bwd [{os.environ.get("DA_ATTN_BWD", "42")}]', bwd, 5 * big)
del Z, dZ, dE1, dE2, E2
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for D_, rows in ((64, B * n * n), (256, B * n)):
    s_, dy = r(rows, D_), r(rows, D_)
    ds, dg, db, gam = torch.empty_like(s_), torch.empty(D_, device=dev), torch.empty(D_, device=dev), torch.ones(D_, device=dev)
    timed(f'layernorm_bwd D={D_} rows={rows}', lambda: L.check(L.lib().da_layernorm_bwd(L.ptr(s_), L.ptr(gam), L.ptr(dy), L.ptr(ds), L.ptr(dg), L.ptr(db),
          L.ptr(ws), ws.numel(), rows, D_, L.stream()), 'ln'), 3 * rows * D_ * 4)
R = B * n * n
for fin, fout in ((64, 256), (256, 64), (64, 128), (128, 64)):
    x, dy = r(R, fin), r(R, fout)
    dW, db = torch.empty(fout, fin, device=dev), torch.empty(fout, device=dev)
    timed(f'linear_wgrad {fin}->{fout}', lambda: L.check(L.lib().da_linear_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dW), L.ptr(db), L.ptr(ws), ws.numel(), R, fin,
          fout, L.stream()), 'wg'), R * (fin + fout) * 4)
    timed(f'  library dY^T X {fin}->{fout}', lambda: dy.t() @ x, R * (fin + fout) * 4, reps=3)
T = r(B, n * n, de)
z = torch.empty(B, 4, de, device=dev); arg = torch.empty(B, 2, de, dtype=torch.int32, device=dev)
timed('pool_stats_fwd E', lambda: L.check(L.lib().da_pool_stats_fwd(L.ptr(T), L.ptr(z), L.ptr(arg), L.ptr(ws), ws.numel(), B, n * n, de, L.stream()), 'p'),
      B * n * n * de * 4)
