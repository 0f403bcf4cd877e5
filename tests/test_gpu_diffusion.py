"""GPU parity: diffusion-side HIP kernels (through the C-ABI) vs the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import load, counter_noise, ATOMS, BONDS

pytestmark = pytest.mark.gpu
KX, KE = len(ATOMS), len(BONDS)
NM, EM = [28, 0], [4, 5, 6]
T = lambda a: torch.from_numpy(np.asarray(a))
D = 'cuda:0'


def _ph(X, E, nm, am=None, mol=None, y=None):
    from diffalign_amd.placeholder import PlaceHolder
    B = X.shape[0]
    return PlaceHolder(X=X.to(D), E=E.to(D), y=(torch.zeros(B, 1) if y is None else y).to(D), node_mask=nm.to(D),
                       atom_map_numbers=None if am is None else am.to(D), mol_assignments=None if mol is None else mol.to(D))


def test_mask_and_collapse_golden():
    z = load('container_masks')
    p = _ph(T(z['in_X']), T(z['in_E']), T(z['node_mask'])).mask()
    assert torch.equal(p.X.cpu(), T(z['out_X'])) and torch.equal(p.E.cpu(), T(z['out_E']))
    c = _ph(T(z['in_X']), T(z['in_E']), T(z['node_mask'])).mask(collapse=True)
    assert np.array_equal(c.X.cpu().numpy(), z['col_X']) and np.array_equal(c.E.cpu().numpy(), z['col_E'])


@pytest.mark.parametrize('tag', ['rag', 'full'])
@pytest.mark.parametrize('mode', ['product_and_sn', 'atom_mapping', 'sn', 'product'])
def test_free_masks_golden(tag, mode):
    from diffalign_amd import graph
    z = load('container_masks')
    X = F.one_hot(T(z[f'{tag}_X_idx']), KX).float(); E = F.one_hot(T(z[f'{tag}_E_idx']), KE).float()
    orig = _ph(X, E, T(z[f'{tag}_node_mask']), T(z[f'{tag}_am']), T(z[f'{tag}_mol']))
    _, mX, mE = graph.apply_mask(orig, orig, list(ATOMS), list(BONDS), mask_nodes=mode, mask_edges=mode,
                                 return_masks=True)
    assert np.array_equal(mX[..., 0].cpu().numpy(), z[f'{tag}_{mode}_mask_X'])
    assert np.array_equal(mE[..., 0].cpu().numpy(), z[f'{tag}_{mode}_mask_E'])


def test_apply_mask_values_golden():
    from diffalign_amd import graph
    z = load('container_masks')
    X = F.one_hot(T(z['rag_X_idx']), KX).float(); E = F.one_hot(T(z['rag_E_idx']), KE).float()
    orig = _ph(X, E, T(z['rag_node_mask']), T(z['rag_am']), T(z['rag_mol']))
    zt = _ph(T(z['apply_zX']), T(z['apply_zE']), T(z['rag_node_mask']), T(z['rag_am']), T(z['rag_mol']))
    for al in (0, 1):
        r = graph.apply_mask(orig, zt, list(ATOMS), list(BONDS), mask_nodes='product_and_sn',
                             mask_edges='product_and_sn', as_logits=bool(al))
        assert torch.equal(r.X.cpu(), T(z[f'apply_{al}_X'])) and torch.equal(r.E.cpu(), T(z[f'apply_{al}_E']))


def _transition(fam):
    from diffalign_amd import noise_schedule as NS
    if fam == 'abs':
        return NS.AbsorbingStateTransitionMaskNoEdge(timesteps=10, x_classes=KX, e_classes=KE, y_classes=1,
                                                     abs_state_position_x=27, abs_state_position_e=0,
                                                     node_idx_to_mask=NM, edge_idx_to_mask=EM, device=D)
    g = torch.Generator().manual_seed(11)
    xm = torch.rand(KX, generator=g); xm = xm / xm.sum()
    em = torch.rand(KE, generator=g); em = em / em.sum()
    return NS.MarginalUniformTransition(xm, em, 1, 'cosine', 10, node_idx_to_mask=NM, edge_idx_to_mask=EM, device=D)


def test_transition_tables_match_golden():
    z = load('transitions')
    for fam, tag in (('abs', 'abs10'), ('mar', 'mar10')):
        tm = _transition(fam)
        for f in ('Qts_x', 'Qts_e', 'Qt_bars_x', 'Qt_bars_e', 'x_limit', 'e_limit'):
            assert (getattr(tm, f).cpu() - T(z[f'{tag}_{f}'])).abs().max() <= 1e-6


@pytest.mark.parametrize('fam', ['abs', 'mar'])
@pytest.mark.parametrize('t', [10, 5, 1])
def test_posterior_probs_and_draws_golden(fam, t):
    from diffalign_amd import helpers
    z = load('posterior_draws')
    tm = _transition(fam)
    k = f'{fam}_t{t}_'
    nm = T(z[f'{fam}_node_mask'])
    B, n = nm.shape
    zt = _ph(F.one_hot(T(z[k + 'zt_X']), KX).float(), F.one_hot(T(z[k + 'zt_E']), KE).float(), nm)
    pred = _ph(T(z[k + 'logit_X']), T(z[k + 'logit_E']), nm)      # raw logits: the kernel applies log_softmax*T
    t_arr = torch.full((B, 1), t, dtype=torch.long, device=D)
    prob = helpers.get_p_zs_given_zt(tm, t_arr, pred, zt, return_prob=True)
    # fp32 tolerance: 2e-6 absolute on probabilities
    assert (prob.X.cpu() - T(z[k + 'prob_X'])).abs().max() <= 2e-6
    assert (prob.E.cpu() - T(z[k + 'prob_E'])).abs().max() <= 2e-6
    # standalone sampler: bit-exact given the reference's probabilities and the same noise
    c = int(z[k + 'noise_counter'])
    qX = counter_noise((B * n, KX), c).reshape(B, n, KX).to(D)
    qE = counter_noise((B * n * n, KE), c + 1).reshape(B, n, n, KE).to(D)
    pg = _ph(T(z[k + 'prob_X']), T(z[k + 'prob_E']), nm)
    _, Xs, Es = helpers.sample_discrete_features(pg, noise=(qX, qE), return_indices=True)
    assert np.array_equal(Xs.cpu().numpy(), z[k + 'zs_X']) and np.array_equal(Es.cpu().numpy(), z[k + 'zs_E'])
    # fused kernel draws: equal to the reference's except where the race is a near-tie
    _, _, zsX, zsE = helpers.posterior_sample(pred.X, pred.E, zt, tm.get_Qt(t_arr), tm.get_Qt_bar(t_arr - 1),
                                              tm.get_Qt_bar(t_arr), noise=(qX, qE))
    mism_X = (zsX.argmax(-1).cpu().numpy() != z[k + 'zs_X']).mean()
    mism_E = (zsE.argmax(-1).cpu().numpy() != z[k + 'zs_E']).mean()
    assert mism_X <= 1e-3 and mism_E <= 1e-4, (mism_X, mism_E)
    assert torch.equal(zsE, zsE.transpose(1, 2))
    # the draw-only call (tile kernel: one evaluation per unordered pair) against the call that also returns probabilities
    _, _, zsX2, zsE2 = helpers.posterior_sample(pred.X, pred.E, zt, tm.get_Qt(t_arr), tm.get_Qt_bar(t_arr - 1),
                                                tm.get_Qt_bar(t_arr), noise=(qX, qE), want_prob=True)
    assert torch.equal(zsX, zsX2) and torch.equal(zsE, zsE2)


def test_fused_step_vs_oracle_full_size_properties():
    """BASELINE-size shapes (n = 300): size-independent properties + oracle agreement on a slice."""
    from diffalign_amd import helpers, graph, synthetic as S
    B, n = 4, 300
    true = S.make_reaction_batch(B, n, seed=7, P=142)
    zt = S.noised_state(true, t=50, T=100, abs_x=27, seed=8)
    g = torch.Generator().manual_seed(9)
    lX = torch.randn(B, n, KX, generator=g) * 2; lE = torch.randn(B, n, n, KE, generator=g) * 2
    qX = S.exp_noise((B, n, KX), 10); qE = S.exp_noise((B, n, n, KE), 11)
    tabs = O.absorbing_tables(100, KX, KE, 1, 27, 0, NM, EM)
    mx, me = O.free_masks(true['X'], true['node_mask'], true['atom_map_numbers'], 'product_and_sn', 28)
    pX, pE, zX, zE = O.reverse_step(lX, lE, zt['X'], zt['E'], zt['node_mask'], tabs, 50, true['X'], true['E'], mx, me,
                                    qX, qE)
    from diffalign_amd import noise_schedule as NS
    tm = NS.AbsorbingStateTransitionMaskNoEdge(100, KX, KE, 1, abs_state_position_x=27, abs_state_position_e=0,
                                               node_idx_to_mask=NM, edge_idx_to_mask=EM, device=D)
    orig = _ph(true['X'], true['E'], true['node_mask'], true['atom_map_numbers'], true['mol_assignments'])
    z = _ph(zt['X'], zt['E'], zt['node_mask'], true['atom_map_numbers'], true['mol_assignments'])
    masks = graph.get_node_masks(orig, list(ATOMS), list(BONDS), 'product_and_sn', 'product_and_sn')
    assert torch.equal(masks[0].cpu(), mx) and torch.equal(masks[1].cpu(), me)
    t_arr = torch.full((B, 1), 50, dtype=torch.long, device=D)
    gpX, gpE, gzX, gzE = helpers.posterior_sample(lX.to(D), lE.to(D), z, tm.get_Qt(t_arr), tm.get_Qt_bar(t_arr - 1),
                                                  tm.get_Qt_bar(t_arr), noise=(qX.to(D), qE.to(D)), orig=orig,
                                                  node_masks=masks, want_prob=True)
    assert (gpX.cpu() - pX).abs().max() <= 2e-6 and (gpE.cpu() - pE).abs().max() <= 2e-6
    _, _, tzX, tzE = helpers.posterior_sample(lX.to(D), lE.to(D), z, tm.get_Qt(t_arr), tm.get_Qt_bar(t_arr - 1),
                                              tm.get_Qt_bar(t_arr), noise=(qX.to(D), qE.to(D)), orig=orig, node_masks=masks)
    assert torch.equal(tzX, gzX) and torch.equal(tzE, gzE)         # draw-only tile kernel == per-pair kernel, bit for bit
    assert (gzX.cpu().argmax(-1) != zX.argmax(-1)).float().mean() <= 1e-3
    assert (gzE.cpu().argmax(-1) != zE.argmax(-1)).float().mean() <= 1e-4
    # properties: one-hot rows, symmetric E, condition (product side) untouched, idempotent apply_mask
    assert torch.all(gzX.sum(-1) == 1) and torch.all(gzE.sum(-1) == 1)
    assert torch.equal(gzE, gzE.transpose(1, 2))
    fixed = ~mx
    assert torch.equal(gzX.cpu()[fixed], true['X'][fixed])
    again = graph.apply_mask(orig, _ph(gzX.cpu(), gzE.cpu(), true['node_mask']), list(ATOMS), list(BONDS),
                             node_masks=masks)
    assert torch.equal(again.X, gzX) and torch.equal(again.E, gzE)


def test_empty_batch_and_bad_args():
    from diffalign_amd import _lib as L
    from diffalign_amd.placeholder import PlaceHolder
    p = PlaceHolder(X=torch.zeros(0, 5, 3, device=D), E=torch.zeros(0, 5, 5, 2, device=D), y=torch.zeros(0, 1, device=D),
                    node_mask=torch.zeros(0, 5, dtype=torch.bool, device=D)).mask()
    assert p.X.shape == (0, 5, 3)
    with pytest.raises(L.DiffAlignHipError):
        PlaceHolder(X=torch.zeros(1, 2, 3), E=torch.zeros(1, 2, 2, 2), y=torch.zeros(1, 1),
                    node_mask=torch.ones(1, 2, dtype=torch.bool)).mask()     # CPU tensors: no fallback


def test_step_tables_equal_gathered_tables():
    """DiffusionTransition.step_tables (one step number for the whole batch: slices of tables laid out once per batch size, or built
    inline beyond the memory cap) == get_Qt / get_Qt_bar on a constant t array."""
    from diffalign_amd.noise_schedule import DiffusionTransition
    for fam in ('abs', 'mar'):
        tm = _transition(fam)
        for cap in (DiffusionTransition.STEP_TABLE_BYTES_MAX, 0):
            tm.STEP_TABLE_BYTES_MAX = cap
            for B in (5, 3):
                for t in (10, 9, 3, 1):
                    Qt, Qsb, Qtb = tm.step_tables(t, B)
                    t_arr = torch.full((B, 1), t, dtype=torch.long, device=D)
                    for got, want in ((Qt, tm.get_Qt(t_arr)), (Qsb, tm.get_Qt_bar(t_arr - 1)), (Qtb, tm.get_Qt_bar(t_arr))):
                        assert torch.equal(got.X, want.X) and torch.equal(got.E, want.E)
                        assert got.X.is_contiguous() and got.E.is_contiguous()
