"""Micro-benchmark of the node chain between two edge kernels (da_x_post, da_qkv_frag): python tools/bench_node_chain.py [B] [n]
Prints the time per launch (HIP events over 50 back-to-back launches) and the error against fp32 torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffalign_amd import pack, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dx = dff = 256
D = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
lib = L.lib()
R = B * n
S = L.stream()


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rnd = lambda *s: torch.randn(*s, generator=g)
p = 'tf_layers.0'
sd = {p + '.self_attn.x_out.weight': rnd(dx, dx) * 0.06, p + '.self_attn.x_out.bias': rnd(dx) * 0.1,
      p + '.linX1.weight': rnd(dff, dx) * 0.06, p + '.linX1.bias': rnd(dff) * 0.1,
      p + '.linX2.weight': rnd(dx, dff) * 0.06, p + '.linX2.bias': rnd(dx) * 0.1,
      p + '.normX1.weight': 1 + 0.1 * rnd(dx), p + '.normX1.bias': 0.1 * rnd(dx),
      p + '.normX2.weight': 1 + 0.1 * rnd(dx), p + '.normX2.bias': 0.1 * rnd(dx)}
blob = pack.x_post_blob(sd, p).to(D)
newX_in, X = rnd(R, dx).to(D), rnd(R, dx).to(D)
mask = (torch.rand(R, generator=g) < 0.9).to(torch.uint8).to(D)
out = torch.empty(R, dx, device=D)
call = lambda: L.check(lib.da_x_post(L.ptr(newX_in), L.ptr(X), L.ptr(mask), L.ptr(blob), L.ptr(out), R, dx, dff, S), 'da_x_post')
us = timed(call)
w = {k[len(p) + 1:]: v.to(D) for k, v in sd.items()}
nx = F.linear(newX_in, w['self_attn.x_out.weight'], w['self_attn.x_out.bias']) * mask[:, None].float()
x1 = F.layer_norm(X + nx, (dx,), w['normX1.weight'], w['normX1.bias'])
ref = F.layer_norm(x1 + F.linear(F.linear(x1, w['linX1.weight'], w['linX1.bias']).relu(), w['linX2.weight'], w['linX2.bias']), (dx,),
                   w['normX2.weight'], w['normX2.bias'])
print(f'da_x_post   B={B} n={n} rows={R}: {us:6.1f} us   rel err {((out - ref).norm() / ref.norm()).item():.2e}')
blob16 = pack.x_post16_blob(sd, p).to(D)
assert blob16.numel() == lib.da_x_post16_blob_bytes(dx, dff)
out16 = torch.empty(R, dx, device=D)
call = lambda: L.check(lib.da_x_post16(L.ptr(newX_in), L.ptr(X), L.ptr(mask), L.ptr(blob16), L.ptr(out16), R, dx, dff, S), 'da_x_post16')
us = timed(call)
print(f'da_x_post16 B={B} n={n} rows={R}: {us:6.1f} us   rel err {((out16 - ref).norm() / ref.norm()).item():.2e}   vs 32-row kernel {((out16 - out).norm() / ref.norm()).item():.2e}')

Wq = torch.cat([rnd(dx, dx) * 0.06 for _ in range(3)])
bq = (rnd(3 * dx) * 0.1).to(D)
Wp = pack.bf16_weight(Wq).to(D)
NT = (n + 31) // 32
Q = torch.zeros(R, dx, device=D)
kvf = torch.zeros(lib.da_kv_frag_bytes(B, n, dx), dtype=torch.uint8, device=D)
call = lambda: L.check(lib.da_qkv_frag(L.ptr(X), L.ptr(Wp), L.ptr(bq), L.ptr(mask), L.ptr(Q), L.ptr(kvf), B, n, dx, S), 'da_qkv_frag')
us = timed(call)
qref = F.linear(X, Wq[:dx].to(D), bq[:dx]) * mask[:, None].float()
import hashlib
print(f'da_qkv_frag B={B} n={n} rows={R}: {us:6.1f} us   rel err(Q) {((Q - qref).norm() / qref.norm()).item():.2e}   '
      f'kv words sha1 {hashlib.sha1(kvf.cpu().numpy().tobytes()).hexdigest()[:12]}')
