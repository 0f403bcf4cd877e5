"""Per-segment cycle totals of the edge-layer kernel (diagnostic build: tools/build_variant.sh STAMP k_edge_layer.hip -DEXP_STAMP).
DA_LIB=diffalign_amd/libdiffalign_hip_STAMP.so python tools/stamp_edge.py [B] [n]"""
import os, sys, ctypes
sys.argv = [sys.argv[0]] + (sys.argv[1:3] if len(sys.argv) >= 3 else ['64', '300']) + ['3']
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_edge.py')).read())
import numpy as np
NS, NWAVES = 12, 2048
buf = (ctypes.c_longlong * (NWAVES * NS))()
lib = L.lib()
rc = lib.da_debug_read_stamps(buf, NWAVES * NS)
a = np.frombuffer(buf, dtype=np.int64).reshape(NWAVES, NS).astype(np.float64)
names = ['row setup', 'tile head+GEMM1(0)', 'block loop rest', 'resid+LN1', 'FFN', 'LN2+store', 'row end', 'blk: loop1', 'blk: loop2', 'blk: T+GEMM2', '-']
rt = a[:, 11].copy(); a[:, 11] = 0; tot = a.sum(1)
tiles = B * n * ((n + 31) // 32) / NWAVES
print(f'rc={rc} waves {NWAVES}; cycles/wave total mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}; tiles/wave {tiles:.1f}')
for i, nm_ in enumerate(names[:10]):
    print(f'  {nm_:22s} {a[:, i].mean() / tiles:9.0f} cyc/tile  {100 * a[:, i].mean() / tot.mean():5.1f} %')
print(f'realtime ticks (100 MHz) per wave: mean {rt.mean():.0f} max {rt.max():.0f} -> in-kernel clock {tot.mean() / rt.mean() * 100:.0f} MHz')
w = tot.reshape(256, 8)
print('by XCD (blockIdx % 8):', np.round(w.reshape(32, 8, 8).mean((0, 2)) / 1e3).astype(int))
print('by wave slot:', np.round(w.mean(0) / 1e3).astype(int))
print('block totals: min %d  p50 %d  max %d k-cycles' % (w.mean(1).min() / 1e3, np.median(w.mean(1)) / 1e3, w.mean(1).max() / 1e3))
