"""Phase cycle counts of da_extra_eigen_solve (EXP_EIG_STAMP build).  DA_LIB=... python tools/stamp_eig.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from diffalign_amd import extra_features as XF, synthetic as S, _lib as L
from tests.golden.common import extra_info, ATOMS
B, n = 64, 300
D = torch.device('cuda:0')
info = extra_info(SimpleNamespace)
true = S.make_reaction_batch(B, n, seed=5, P=142)
mf = XF.ExtraMolecularFeatures(info)
z = S.noised_state(true, 250, 500, len(ATOMS) - 2, seed=6)
E, nm = z['E'].to(D), z['node_mask'].to(D)
if 'dense' in sys.argv:        # what a random-weight denoiser leaves on the free side (the benchmark's states): a dense random graph
    g = torch.Generator().manual_seed(9)
    free = (true['X'].argmax(-1) != len(ATOMS) - 1).cumprod(-1).bool()          # nodes before the supernode
    cls = torch.randint(0, 4, (B, n, n), generator=g)
    cls = torch.triu(cls, 1); cls = cls + cls.transpose(1, 2)
    pair = free[:, :, None] & free[:, None, :]
    Ec = z['E'].argmax(-1)
    Ec = torch.where(pair, cls, Ec)
    E = torch.nn.functional.one_hot(Ec, z['E'].shape[-1]).float().to(D)
nm8 = nm.view(torch.uint8)
A, Lap, vw, vp = XF._adjacency(E, nm8, mf.consts(D)[1], True)
pn = torch.randn(B, n, device=D)
out = torch.zeros(B, 6, dtype=torch.float64, device=D)
nc = torch.empty(B, device=D); ev5 = torch.empty(B, 5, device=D); nl = torch.empty(B, n, device=D); v2 = torch.empty(B, n, 2, device=D)
if 'cached' in sys.argv:      # second call of a chain: the fixed block comes back from the cache
    from diffalign_amd import graph
    from diffalign_amd.placeholder import PlaceHolder
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    masks = graph.get_node_masks(data, list(ATOMS), ['none', 'SINGLE', 'DOUBLE', 'TRIPLE', 'mol', 'within', 'across'], 'product_and_sn', 'product_and_sn')
    cache = XF.EigCache.from_free_masks(masks[1], data.node_mask)
    for _ in range(3):
        Lc = Lap.clone()
        L.check(L.lib().da_extra_eigen_solve_cached(L.ptr(Lc), L.ptr(nm8), L.ptr(pn), L.ptr(nc), L.ptr(ev5), L.ptr(nl), L.ptr(v2), L.ptr(out), C.c_int(B), C.c_int(n),
                                                    L.ptr(cache.fixed_from), L.ptr(cache.state), L.ptr(cache.store), C.c_int(cache.mmax), L.stream()), 'x')
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    Lcs = [Lap.clone() for _ in range(5)]
    ev0.record()
    for Lc in Lcs:
        L.check(L.lib().da_extra_eigen_solve_cached(L.ptr(Lc), L.ptr(nm8), L.ptr(pn), L.ptr(nc), L.ptr(ev5), L.ptr(nl), L.ptr(v2), L.ptr(out), C.c_int(B), C.c_int(n),
                                                    L.ptr(cache.fixed_from), L.ptr(cache.state), L.ptr(cache.store), C.c_int(cache.mmax), L.stream()), 'x')
    ev1.record(); torch.cuda.synchronize()
    print('cached call: %.3f ms per launch; states' % (ev0.elapsed_time(ev1) / 5), cache.state.unique().tolist())
for _ in range(0 if 'cached' in sys.argv else 2):
    Lc = Lap.clone()
    L.check(L.lib().da_extra_eigen_solve(L.ptr(Lc), L.ptr(nm8), L.ptr(pn), L.ptr(nc), L.ptr(ev5), L.ptr(nl), L.ptr(v2), L.ptr(out), C.c_int(B), C.c_int(n), L.stream()), 'x')
torch.cuda.synchronize()
o = out.cpu()
print('mean k-cycles per graph: tridiag %.0f  eigenvalues %.0f  inverse iteration %.0f  back-transform %.0f' % tuple((o[:, :4].mean(0) / 1e3).tolist()))
print('max  k-cycles per graph: tridiag %.0f  eigenvalues %.0f  inverse iteration %.0f  back-transform %.0f' % tuple((o[:, :4].max(0).values / 1e3).tolist()))

try:
    buf = (C.c_longlong * 8)()
    L.lib().da_debug_read_eig_phases.argtypes = [C.c_void_p]
    if L.lib().da_debug_read_eig_phases(buf) == 0:
        calls = 8 if 'cached' in sys.argv else 2
        print('Householder step phases of graph 0, k-cycles per call: ' + '  '.join(f'({chr(97 + i)}) {buf[i] / calls / 1e3:.0f}' for i in range(6)))
except AttributeError:
    pass
