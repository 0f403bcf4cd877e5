"""Outputs of the round-2 eigensolver (k_extra_eig_solve as of commit 082dba9) on seeded synthetic chains, recorded ON THE GPU BOX with
that build: the round-3 kernel (LDS-resident triangle, segmented Sturm counts and tridiagonal solves) must reproduce every feature bit
for bit (tests/test_gpu_extra_features.py::test_eigensolver_matches_round2_outputs).  Not reference data: a regression pin of this
library against itself.   python tests/golden/make_eig_r2.py OUT.npz"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

CASES = [(300, 142, 3), (300, None, 4), (72, 28, 5), (24, 4, 4), (140, 60, 2)]
STEPS = (480, 300, 120, 10, 0)


def run_case(n, P, B, cached):
    from tests.golden.common import extra_info, ATOMS
    from diffalign_amd import extra_features as XF, graph, synthetic as S
    from diffalign_amd.placeholder import PlaceHolder
    D = 'cuda:0'
    info = extra_info(SimpleNamespace)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    true = S.make_reaction_batch(B, n, seed=11, P=P, kx=len(ATOMS))
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    bonds = ['none', 'SINGLE', 'DOUBLE', 'TRIPLE', 'mol', 'within', 'across']
    masks = graph.get_node_masks(data, list(ATOMS), bonds, 'product_and_sn', 'product_and_sn')
    cache = XF.EigCache.from_free_masks(masks[1], data.node_mask) if cached else None
    pn = torch.randn(B, n, generator=torch.Generator().manual_seed(3)).to(D)
    out = []
    for step, t in enumerate(STEPS):
        z = S.noised_state(true, t, 500, len(ATOMS) - 2, seed=20 + step)
        X, E, nm = z['X'].to(D), z['E'].to(D), z['node_mask'].to(D)
        fx, fy = XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn, eig_cache=cache)
        out.append((fx.cpu().numpy(), fy.cpu().numpy()))
    return out


if __name__ == '__main__':
    res = {}
    for ci, (n, P, B) in enumerate(CASES):
        for cached in (0, 1):
            for si, (fx, fy) in enumerate(run_case(n, P, B, bool(cached))):
                res[f'c{ci}_k{cached}_s{si}_x'] = fx; res[f'c{ci}_k{cached}_s{si}_y'] = fy
    np.savez_compressed(sys.argv[1], **res)
    print('wrote', sys.argv[1], len(res), 'arrays')
