"""Stage times of the y chain (EXP_Y_STAMP build of k_node.hip): DA_LIB=.../libdiffalign_hip_ystamp.so python tools/stamp_y.py [B] [n]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import FULL, make_cfg, make_info
from diffalign_amd import engine, _lib as L
from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = DiscreteDenoisingDiffusion(make_cfg(FULL), make_info(), device=dev).eval()
pm = model.model._pack(dev)
lw, nxt = pm.layers[0], pm.layers[1]
y = torch.randn(B, pm.dy, device=dev); xy = torch.randn(B, pm.dy, device=dev)
st = torch.randn(B, engine.tail_chunks(n), 4, pm.de, device=dev).abs()
out = torch.empty(B, pm.dy, device=dev); film = engine.film_buffers(pm, B, dev)
names = ['pooled Etoy stats + y load', 'e_y.lin (256 -> 64)', 'sum, y_out.0, y_out.2', 'residual + norm_y1', 'lin_y1 (64 -> 2048)', 'lin_y2 (2048 -> 64)',
         'residual + norm_y2 + store', 'four FiLM projections', 'e_out bias fold (256 -> 64)']
acc = [0.0] * 9
for it in range(12):
    engine.y_update_film(pm, lw, nxt, y, xy, st, B, n, out, film)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    L.lib().da_debug_read_ystamps(buf)
    if it >= 2:
        for i in range(9): acc[i] += (buf[i + 1] - buf[i]) * 10.0 / 1e3 / 10
for nm, v in zip(names, acc): print(f'{nm:32s} {v:6.2f} us')
print(f'{"total (workgroup 0)":32s} {sum(acc):6.2f} us')
