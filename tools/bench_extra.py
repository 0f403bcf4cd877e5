"""Cost of the on-device extra features at the benchmark's size.  python tools/bench_extra.py [B] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from diffalign_amd import extra_features as XF, synthetic as S, _lib as L
from tests.golden.common import extra_info, ATOMS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
D = torch.device('cuda:0')
info = extra_info(SimpleNamespace)
true = S.make_reaction_batch(B, n, seed=5, P=(n - 16) // 2)
ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3
for t in (50, 250, 450):
    z = S.noised_state(true, t, 500, len(ATOMS) - 2, seed=6)
    X, E, nm = z['X'].to(D), z['E'].to(D), z['node_mask'].to(D)
    pn = torch.randn(B, n, device=D)
    nm8 = nm.view(torch.uint8)
    bo = mf.consts(D)[1]
    A, Lap, vw, vp = XF._adjacency(E, nm8, bo, True)
    print(f't={t}: total {timed(lambda: XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn)):.2f} ms | adjacency {timed(lambda: XF._adjacency(E, nm8, bo, True)):.3f}'
          f' | cycles (2 bmm + kernel) {timed(lambda: XF._cycles(A, nm8)):.3f} | eigh fp64 {timed(lambda: torch.linalg.eigh(Lap)):.2f}'
          f' | eigen via torch.linalg.eigh {timed(lambda: XF._eigen(Lap.clone(), nm8, pn, solver="torch")):.2f} | eigen via da_extra_eigen_solve {timed(lambda: XF._eigen(Lap.clone(), nm8, pn, solver="hip")):.2f} ms', flush=True)
