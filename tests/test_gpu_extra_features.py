"""GPU: extra input features on the device (csrc/k_extra.hip through the C-ABI + torch.bmm / torch.linalg.eigh) against the
reference's outputs (tests/golden/extra_features.npz) and the CPU oracle.  Tolerance: cycle counts / eigenvalue features
2e-5 absolute (fp32 features of fp64 eigenvalues), charge / valency exact; the eigenvector features are compared through
basis-independent properties only (see tests/golden/common.py:check_extra)."""
from types import SimpleNamespace

import os

import pytest
import torch

from oracle import oracle as O
from tests.golden.common import load, extra_info, extra_case, check_extra, EXTRA_CASES, ATOMS

pytestmark = pytest.mark.gpu
D = torch.device('cuda:0')


@pytest.mark.parametrize('case', EXTRA_CASES)
def test_extra_features_vs_reference_golden(case):
    from diffalign_amd import extra_features as XF
    z = load('extra_features')
    info = extra_info(SimpleNamespace)
    X, E, nm, pad_noise, _, _ = extra_case(z, case)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    Xe, ye = XF.compute_extra_features(X.to(D), E.to(D), nm.to(D), ef, mf, pad_noise=pad_noise.to(D))
    torch.cuda.synchronize()
    Lap = O.eigen_features(E, nm, pad_noise)[4]
    check_extra(Xe, ye, case, z, E, nm, Lap)
    # the two calculators on their own (the reference's class interface) give the same columns
    Xr, Er, yr = ef(E.to(D), nm.to(D), pad_noise=pad_noise.to(D))
    Xm, Em, ym = mf(X.to(D), E.to(D))
    assert Er.shape == (*E.shape[:-1], 0) and Em.shape == (*E.shape[:-1], 0)
    assert torch.equal(torch.cat((Xr, Xm), -1)[..., [0, 1, 2, 6, 7]], Xe[..., [0, 1, 2, 6, 7]])
    assert torch.equal(torch.hstack((yr, ym))[:, :5], ye[:, :5]) and torch.equal(ym[:, 0], ye[:, 11])
    # Laplacian written by the kernel == the oracle's, bit for bit (small integers in fp64)
    _, Lk, _, _ = XF._adjacency(E.to(D), nm.to(D).view(torch.uint8), torch.zeros(E.shape[-1], device=D), True)
    assert torch.equal(Lk.cpu(), Lap)


def test_extra_features_cycles_modes_and_bench_size():
    """'cycles' / 'eigenvalues' modes return the reference's column subsets; the benchmark's size (n = 300) against the
    oracle on one graph."""
    from diffalign_amd import extra_features as XF
    from diffalign_amd import synthetic as S
    info = extra_info(SimpleNamespace)
    true = S.make_reaction_batch(2, 300, seed=5, P=142)
    z = S.noised_state(true, 250, 500, len(ATOMS) - 2, seed=6)
    X, E, nm = z['X'], z['E'], z['node_mask']
    pad_noise = torch.randn(nm.shape, generator=torch.Generator().manual_seed(1))
    Xo, yo = O.extra_features(X, E, nm, info.max_n_nodes, info.valencies, info.bond_orders, info.atom_weights.values(), info.max_weight,
                              pad_noise)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    Xe, ye = XF.compute_extra_features(X.to(D), E.to(D), nm.to(D), ef, mf, pad_noise=pad_noise.to(D))
    torch.testing.assert_close(Xe.cpu()[..., :3], Xo[..., :3], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(Xe.cpu()[..., 6:], Xo[..., 6:], atol=0, rtol=0)
    torch.testing.assert_close(ye.cpu(), yo, atol=2e-5, rtol=1e-5)
    xc, Ec, yc = XF.ExtraFeatures('cycles', info)(E.to(D), nm.to(D))
    assert xc.shape == (2, 300, 3) and yc.shape == (2, 5) and torch.equal(yc, ye[:, :5]) and torch.equal(xc, Xe[..., :3])
    xv, Ev, yv = XF.ExtraFeatures('eigenvalues', info)(E.to(D), nm.to(D))
    assert yv.shape == (2, 11) and torch.equal(yv, ye[:, :11])


def test_model_forward_and_sampling_with_extra_features():
    """cfg.neuralnet.extra_features=True: DiscreteDenoisingDiffusion computes the features itself on the device (no
    extra_features_fn, no host round trip): X grows by 8 columns, y by 12; the logits equal the CPU oracle's denoiser fed the
    same features (rel-L2 <= 2e-2); a short reverse chain runs end to end and returns valid symmetric one-hot graphs."""
    from types import SimpleNamespace as ns
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    from tests.golden.common import tiny_dims, init_state_dict, BONDS
    from tests.test_gpu_sampling import make_cfg, make_info
    info = make_info()
    for k, v in vars(extra_info(ns)).items():
        setattr(info, k, v)
    dims = tiny_dims()
    model = DiscreteDenoisingDiffusion(make_cfg(dims, extra=True), info, device='cuda:0').eval()
    sd = init_state_dict({k: v.shape for k, v in model.model.state_dict().items()}, 5)
    model.model.load_state_dict(sd)
    assert sd['mlp_in_X.0.weight'].shape[1] == len(ATOMS) + 8 and sd['mlp_in_y.0.weight'].shape[1] == 1 + 12
    true = S.make_reaction_batch(3, 40, seed=9)
    z = S.noised_state(true, 5, 10, len(ATOMS) - 2, seed=10)
    pos = S.shared_pos_encodings(true, dims['hidden_dims']['dx'], seed=4)
    model.pos_encodings_fn = lambda zt: pos.to(D)
    zt = PlaceHolder(X=z['X'].to(D), E=z['E'].to(D), y=z['y'].to(D), node_mask=z['node_mask'].to(D),
                     atom_map_numbers=z['atom_map_numbers'].to(D), mol_assignments=z['mol_assignments'].to(D))
    pn = torch.randn(3, 40, generator=torch.Generator().manual_seed(2)).to(D)
    ext = model.compute_extra_data(zt, pad_noise=pn)
    assert ext.X.shape[-1] == len(ATOMS) + 8 and ext.y.shape[-1] == 13
    model.extra_features_fn = lambda zz: (ext.X[..., len(ATOMS):], ext.y[..., 1:])      # same features, fed by hand
    with torch.no_grad():
        pred = model.forward(zt, pos_encodings=pos.to(D))
    oX, oE, _ = O.denoiser_forward(sd, ext.X.cpu(), ext.E.cpu(), ext.y.cpu(), z['node_mask'], dims['n_layers'], dims['hidden_dims']['n_head'],
                                   (len(ATOMS), len(BONDS), 1), atom_map_numbers=z['atom_map_numbers'], pos_encodings=pos,
                                   mol_assignments=z['mol_assignments'], p_to_r_skip_connection=True)
    rel = lambda a, b: ((a.cpu().double() - b.double()).norm() / b.double().norm()).item()
    assert rel(pred.X, oX) <= 2e-2 and rel(pred.E, oE) <= 2e-2, (rel(pred.X, oX), rel(pred.E, oE))
    model.extra_features_fn = None                                                    # built-in path inside the loop
    data = PlaceHolder(X=true['X'].to(D), E=true['E'].to(D), y=torch.zeros(3, 1, device=D), node_mask=true['node_mask'].to(D),
                       atom_map_numbers=true['atom_map_numbers'].to(D), mol_assignments=true['mol_assignments'].to(D))
    out = model.sample_one_batch(data=data)
    assert torch.equal(out.E, out.E.transpose(1, 2)) and bool((out.X.sum(-1) == 1).all()) and bool((out.E.sum(-1) == 1).all())
    # the chain keeps the fixed side's share of the eigensolver between steps (EigCache): the same seeds give the same samples, bit for
    # bit, with the cache switched off
    from diffalign_amd import extra_features as XF
    seeds = torch.arange(3, device=D) + 77
    with_cache = model.sample_one_batch(data=data, chain_seeds=seeds)
    XF.EIG_CACHE = False
    try:
        without = model.sample_one_batch(data=data, chain_seeds=seeds)
    finally:
        XF.EIG_CACHE = True
    assert torch.equal(with_cache.X, without.X) and torch.equal(with_cache.E, without.E)


def _solve(Lap, nm, pn):
    import ctypes as C
    from diffalign_amd import _lib as L
    B, n, _ = Lap.shape
    Ld = Lap.to(D).contiguous().clone()
    nm8, pnd = nm.to(D).view(torch.uint8).contiguous(), pn.to(D).contiguous()     # (kept alive until the launch has been enqueued)
    ev6 = torch.zeros(B, 6, dtype=torch.float64, device=D)
    nc = torch.empty(B, device=D); ev5 = torch.empty(B, 5, device=D); nl = torch.empty(B, n, device=D); v2 = torch.empty(B, n, 2, device=D)
    L.check(L.lib().da_extra_eigen_solve(L.ptr(Ld), L.ptr(nm8), L.ptr(pnd), L.ptr(nc), L.ptr(ev5),
                                         L.ptr(nl), L.ptr(v2), L.ptr(ev6), C.c_int(B), C.c_int(n), L.stream()), 'da_extra_eigen_solve')
    torch.cuda.synchronize()
    return nc.cpu(), ev5.cpu(), nl.cpu(), v2.cpu(), ev6.cpu()


@pytest.mark.parametrize('n,p,B', [(1, 0.0, 2), (2, 1.0, 2), (3, 0.5, 3), (37, 0.08, 4), (64, 0.6, 3), (300, 0.01, 3), (300, 0.2, 2), (513, 0.004, 1)])
def test_partial_eigensolver_vs_lapack(n, p, B):
    """da_extra_eigen_solve (one launch: Householder tridiagonalisation, Sturm multisection, inverse iteration) against
    torch.linalg.eigh on the CPU (LAPACK, fp64): the count below the threshold, eigenvalue 0 and the 5 after the zero ones to
    1e-9 * ||L||, and the two returned eigenvectors as eigenvectors (residual 1e-8 * ||L||, unit norm)."""
    g = torch.Generator().manual_seed(1000 + n)
    nm = torch.ones(B, n, dtype=torch.bool)
    if n > 8:
        nm[0, n - n // 5:] = False                                  # one graph with padding
    A = (torch.rand(B, n, n, generator=g) < p).double(); A = torch.triu(A, 1); A = A + A.transpose(1, 2)
    A = A * nm[:, :, None] * nm[:, None, :]
    Lap = torch.diag_embed(A.sum(-1)) - A
    Lap = Lap + torch.diag_embed((~nm).double() * 2 * n)
    pn = torch.randn(B, n, generator=g)
    nc, ev5, nl, v2, ev6 = _solve(Lap, nm, pn)
    w, _ = torch.linalg.eigh(Lap)
    nreal = nm.sum(1).double()
    scale = max(1.0, float(Lap.abs().sum(-1).max()))
    for b in range(B):
        ncomp = int((w[b] / nreal[b] < 1e-5).sum())
        assert int(nc[b]) == ncomp
        assert abs(float(ev6[b, 0] - w[b, 0])) <= 1e-9 * scale
        for k in range(5):
            if ncomp + k < n:
                assert abs(float(ev6[b, 1 + k] - w[b, ncomp + k])) <= 1e-9 * scale, (b, k)
                assert abs(float(ev5[b, k]) - float(w[b, ncomp + k] / nreal[b])) <= 1e-6 * max(1.0, float(w[b, ncomp + k] / nreal[b]))
            else:
                assert float(ev5[b, k]) == 2.0
        for k in range(2):
            col = ncomp + k
            if col < n and bool(nm[b, col]):
                vfull = v2[b, :, k].double()
                if bool(nm[b].all()):                                   # (masked rows are zeroed in the feature, not in the vector)
                    assert abs(float(vfull.norm()) - 1) <= 1e-5
                    assert float((Lap[b] @ vfull - w[b, col] * vfull).abs().max()) <= 1e-5 * scale, (b, k)
        assert (nl[b][~nm[b]] == 0).all()


@pytest.mark.parametrize('n,reals', [(1, [1, 1]), (2, [2, 1]), (5, [5, 3, 0 + 1])])
def test_extra_features_tiny_graphs_vs_oracle(n, reals):
    """Degenerate sizes: one or two nodes, graphs that are almost all padding, B = 0."""
    from diffalign_amd import extra_features as XF
    info = extra_info(SimpleNamespace)
    B = len(reals)
    g = torch.Generator().manual_seed(50 + n)
    kx, ke = len(ATOMS), 7
    nm = torch.zeros(B, n, dtype=torch.bool)
    X = torch.zeros(B, n, kx); E = torch.zeros(B, n, n, ke)
    for b, r in enumerate(reals):
        nm[b, :r] = True
        X[b, torch.arange(r), torch.randint(1, kx - 1, (r,), generator=g)] = 1
        cls = torch.randint(0, 4, (r, r), generator=g); cls = torch.triu(cls, 1); cls = cls + cls.t()
        E[b, :r, :r] = torch.nn.functional.one_hot(cls, ke).float()
        E[b, torch.arange(n), torch.arange(n)] = 0
    pn = torch.randn(B, n, generator=g)
    Xo, yo = O.extra_features(X, E, nm, info.max_n_nodes, info.valencies, info.bond_orders, info.atom_weights.values(), info.max_weight, pn)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    for solver in ('hip', 'torch'):
        XF.EIGENSOLVER = solver
        try:
            Xe, ye = XF.compute_extra_features(X.to(D), E.to(D), nm.to(D), ef, mf, pad_noise=pn.to(D))
        finally:
            XF.EIGENSOLVER = 'hip'
        torch.testing.assert_close(Xe.cpu()[..., :3], Xo[..., :3], atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(Xe.cpu()[..., 6:], Xo[..., 6:], atol=0, rtol=0)
        torch.testing.assert_close(ye.cpu(), yo, atol=2e-5, rtol=1e-5)
        assert (Xe.cpu()[..., 3:6][~nm] == 0).all()
    Xe, ye = XF.compute_extra_features(X[:0].to(D), E[:0].to(D), nm[:0].to(D), ef, mf)
    assert Xe.shape == (0, n, 8) and ye.shape == (0, 12)


@pytest.mark.parametrize('n,P,B', [(300, 142, 3), (300, None, 4), (40, 10, 5)])
def test_eigensolver_fixed_block_cache_is_bit_identical(n, P, B):
    """da_extra_eigen_solve_cached (extra_features.EigCache): along a chain the product side and the supernode keep their edges, the
    first call keeps that block's reflectors and the later ones reduce only the rows before it.  Every feature equals the uncached
    solver's bit for bit at every step; a block that shares an edge with the free side is noticed (state -1) and reduced in full."""
    from diffalign_amd import extra_features as XF, graph
    from diffalign_amd import synthetic as S
    from diffalign_amd.placeholder import PlaceHolder
    info = extra_info(SimpleNamespace)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    true = S.make_reaction_batch(B, n, seed=11, P=P, kx=len(ATOMS))
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    bonds = ['none', 'SINGLE', 'DOUBLE', 'TRIPLE', 'mol', 'within', 'across']
    masks = graph.get_node_masks(data, list(ATOMS), bonds, 'product_and_sn', 'product_and_sn')
    cache = XF.EigCache.from_free_masks(masks[1], data.node_mask)
    suno = true['X'].argmax(-1) == len(ATOMS) - 1
    assert torch.equal(cache.fixed_from.cpu().long(), suno.float().argmax(-1)), 'the fixed block starts at the supernode'
    pn = torch.randn(B, n, generator=torch.Generator().manual_seed(3)).to(D)
    for step, t in enumerate((480, 300, 120, 10, 0)):
        z = S.noised_state(true, t, 500, len(ATOMS) - 2, seed=20 + step)
        X, E, nm = z['X'].to(D), z['E'].to(D), z['node_mask'].to(D)
        want = XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn)
        got = XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn, eig_cache=cache)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), f'step {step}'
        assert bool((cache.state == 1).all())
    # a reactant atom bonded to a product atom: the block is not closed, nothing is kept, results unchanged
    cache2 = XF.EigCache.from_free_masks(masks[1], data.node_mask)
    for step, t in enumerate((300, 50)):
        z = S.noised_state(true, t, 500, len(ATOMS) - 2, seed=40 + step)
        E = z['E'].clone()
        last = int(true['node_mask'][0].sum()) - 1
        E[0, 0, last] = 0.; E[0, last, 0] = 0.; E[0, 0, last, 1] = 1.; E[0, last, 0, 1] = 1.
        X, E, nm = z['X'].to(D), E.to(D), z['node_mask'].to(D)
        want = XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn)
        got = XF.compute_extra_features(X, E, nm, ef, mf, pad_noise=pn, eig_cache=cache2)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert int(cache2.state[0]) == -1 and bool((cache2.state[1:] == 1).all())


def test_eigensolver_matches_round2_outputs():
    """Regression pin of the round-3 eigensolver (LDS-resident lower triangle, six barriers per Householder step, Sturm counts and
    tridiagonal solves cut at the exact zeros of the sub-diagonal) against the round-2 kernel's outputs recorded on the GPU
    (tests/golden/eig_r2.npz, make_eig_r2.py): five chains x five steps, with and without the fixed-block cache.  The round-2 kernel
    updated both triangles of the matrix with a contracted a*b + c*d whose rounding depends on the operand order, so the two are not
    bit-comparable in fp64; as fp32 features they agree: counts and cycle features exactly, eigenvalues to 2e-6 relative.  The
    eigenvector features (two vectors, and the component flag taken from the first vector rounded to three decimals) are
    basis-dependent inside degenerate eigenspaces in the reference itself; the same deterministic start keeps them equal here up to
    rounding: compared as shares (printed)."""
    import numpy as np
    from tests.golden.make_eig_r2 import CASES, run_case
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eig_r2.npz'))
    eq = tot = 0
    flag_diff = flag_tot = 0
    vec_bad = vec_tot = vec_skip = 0
    vmax = 0.0
    for ci, (n, P, B) in enumerate(CASES):
        for cached in (0, 1):
            for si, (fx, fy) in enumerate(run_case(n, P, B, bool(cached))):
                wx, wy = z[f'c{ci}_k{cached}_s{si}_x'], z[f'c{ci}_k{cached}_s{si}_y']
                tag = (n, P, B, cached, si)
                assert np.array_equal(fy[:, :6], wy[:, :6]) and np.array_equal(fy[:, 11], wy[:, 11]), tag      # n, cycles, components, weight
                assert np.allclose(fy[:, 6:11], wy[:, 6:11], rtol=2e-6, atol=1e-7), (tag, np.abs(fy[:, 6:11] - wy[:, 6:11]).max())
                assert np.array_equal(fx[..., :3], wx[..., :3]) and np.array_equal(fx[..., 6:], wx[..., 6:]), tag  # cycles, charge / valency
                flag_diff += int((fx[..., 3] != wx[..., 3]).sum()); flag_tot += fx[..., 3].size
                # eigenvectors ncomp, ncomp + 1: column by column up to a sign where their eigenvalues are simple (relative gaps to the
                # neighbours > 1e-4), the projector of the pair where only the two coincide, not compared inside larger clusters
                for g in range(fx.shape[0]):
                    lam = wy[g, 6:9].astype(np.float64)
                    gap = lambda a, b_: abs(a - b_) > 1e-4 * max(abs(a), abs(b_), 1e-30)
                    v, w = fx[g, :, 4:6].astype(np.float64), wx[g, :, 4:6].astype(np.float64)
                    vec_tot += 1
                    if gap(lam[0], lam[1]) and gap(lam[1], lam[2]):
                        d = max(min(np.abs(v[:, k] - w[:, k]).max(), np.abs(v[:, k] + w[:, k]).max()) for k in range(2))
                    elif gap(lam[1], lam[2]):
                        d = np.abs(v @ v.T - w @ w.T).max()
                    elif gap(lam[0], lam[1]):
                        d = min(np.abs(v[:, 0] - w[:, 0]).max(), np.abs(v[:, 0] + w[:, 0]).max())
                    else:
                        vec_skip += 1
                        continue
                    vmax = max(vmax, float(d)); vec_bad += int(d > 1e-4)
                eq += int((fx == wx).sum()) + int((fy == wy).sum()); tot += fx.size + fy.size
    print(f'round-3 eigensolver vs round-2 outputs: {eq / tot:.6f} of all feature values bit-equal; component flag differs on '
          f'{flag_diff} of {flag_tot} nodes; eigenvector pairs: {vec_tot} graphs, {vec_skip} inside larger clusters (not comparable), '
          f'{vec_bad} off by more than 1e-4 up to sign / basis (largest difference {vmax:.3g})')
    assert flag_diff <= 0.005 * flag_tot and vec_bad == 0


@pytest.mark.parametrize('n,B', [(300, 3), (45, 4), (1, 2)])
def test_laplacian_tiles_and_in_place_assembly(n, B):
    """(i) da_extra_adjacency's Laplacian -- diagonal by the row pass, off-diagonal entries by the tiled pass that takes the
    transposed entry of A through LDS -- against the reference's formula (extra_features.py:105-109, :133-146) evaluated with
    torch in fp64, on a NON-symmetric fp32 state (the transposed entry matters) with padded nodes; exact.  (ii) the widened
    denoiser inputs assembled in place (`base=`: the eigensolver writes its six features into the rows itself) equal the
    concatenation of the dense results bit for bit, with and without the fixed-block cache."""
    from diffalign_amd import extra_features as XF, synthetic as S
    g = torch.Generator().manual_seed(n)
    ke = 5
    E = torch.rand(B, n, n, ke, generator=g)
    nm = torch.ones(B, n, dtype=torch.bool)
    if n > 2:
        nm[0, n - 2:] = False; nm[-1, n // 2:] = False
    A, Lap, _, _ = XF._adjacency(E.to(D), nm.to(D).view(torch.uint8), torch.zeros(ke, device=D), True)
    a = torch.zeros(B, n, n)
    for c in range(1, ke):
        a = a + E[..., c]                                       # the kernel's order of additions
    assert torch.equal(A.cpu(), a)
    m2 = (nm[:, :, None] & nm[:, None, :])
    am = (a * m2).double()
    deg = am.sum(-1)
    eye = torch.eye(n, dtype=torch.bool)
    want = torch.where(eye, torch.where(nm, deg - torch.diagonal(a, dim1=1, dim2=2).double(), torch.full_like(deg, 2.0 * n))[:, :, None].expand(B, n, n),
                       -0.5 * (am + am.transpose(1, 2)))
    got = Lap.cpu()
    off = ~eye
    assert torch.equal(got[:, off], want[:, off])
    assert torch.allclose(got[:, eye], want[:, eye], rtol=1e-13, atol=0)     # the row sum's order differs from torch's
    if n < 40:
        return
    info = extra_info(SimpleNamespace)
    ef, mf = XF.ExtraFeatures('all', info), XF.ExtraMolecularFeatures(info)
    true = S.make_reaction_batch(B, n, seed=5, P=142 if n == 300 else None, kx=len(ATOMS))
    z = S.noised_state(true, 200, 500, len(ATOMS) - 2, seed=6)
    X, Ez, nmz = z['X'].to(D), z['E'].to(D), z['node_mask'].to(D)
    y0 = torch.rand(B, 1, generator=g).to(D)
    pn = torch.randn(B, n, generator=g).to(D)
    Xe, ye = XF.compute_extra_features(X, Ez, nmz, ef, mf, pad_noise=pn)
    assert Xe.shape == (B, n, 8) and ye.shape == (B, 12)
    Xw, yw = XF.compute_extra_features(X, Ez, nmz, ef, mf, pad_noise=pn, base=(X, y0))
    assert torch.equal(Xw, torch.cat((X, Xe), -1)) and torch.equal(yw, torch.cat((y0, ye), -1))
    from diffalign_amd import graph
    from diffalign_amd.placeholder import PlaceHolder
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    masks = graph.get_node_masks(data, list(ATOMS), ['none', 'SINGLE', 'DOUBLE', 'TRIPLE', 'mol', 'within', 'across'], 'product_and_sn', 'product_and_sn')
    cache = XF.EigCache.from_free_masks(masks[1], data.node_mask)
    for _ in range(2):
        Xc, yc = XF.compute_extra_features(X, Ez, nmz, ef, mf, pad_noise=pn, eig_cache=cache, base=(X, y0))
        assert torch.equal(Xc, Xw) and torch.equal(yc, yw)
