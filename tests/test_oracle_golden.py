"""Pins oracle/oracle.py (the CPU restatement) to the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import load, weights_of, counter_noise, ATOMS, BONDS

KX, KE = len(ATOMS), len(BONDS)
T = lambda a: torch.from_numpy(np.asarray(a))
NM, EM = [28, 0], [4, 5, 6]


def close(a, b, tol=1e-6):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, err


def test_transition_tables():
    z = load('transitions')
    a = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    m = O.marginal_tables(10, T(z['mar_x_marginals']), T(z['mar_e_marginals']), 1, NM, EM)
    u = O.uniform_tables(10, KX, KE, 1, NM, EM)
    name = dict(Qts_x='Qt_x', Qts_e='Qt_e', Qts_y='Qt_y', Qt_bars_x='Qtb_x', Qt_bars_e='Qtb_e', Qt_bars_y='Qtb_y',
                x_limit='x_limit', e_limit='e_limit', y_limit='y_limit')
    for tag, tab in (('abs10', a), ('mar10', m), ('uni10', u)):
        for ref_name, mine in name.items():
            close(tab[mine], z[f'{tag}_{ref_name}'], 1e-6)
        # the reference's own property tests (tests/test_noise_schedule.py:95-143,237-290)
        for k in ('Qt_x', 'Qt_e', 'Qtb_x', 'Qtb_e'):
            assert (tab[k] >= 0).all()
            close(tab[k].sum(-1), torch.ones_like(tab[k].sum(-1)), 1e-5)
        if tag == 'abs10':
            close(tab['Qtb_x'][0], torch.eye(KX), 1e-6)
    for TT in (100, 500):
        a = O.absorbing_tables(TT, KX, KE, 1, 27, 0, NM, EM)
        sel = z[f'abs{TT}_sel'].tolist()
        for ref_name in ('Qts_x', 'Qts_e', 'Qt_bars_x', 'Qt_bars_e'):
            full = a[name[ref_name]]
            close(full[sel], z[f'abs{TT}_{ref_name}_sel'], 1e-7)
            close(full.double().sum((1, 2)), z[f'abs{TT}_{ref_name}_sum'], 1e-5)
            w = torch.arange(full[0].numel()).reshape(full[0].shape).double()
            close((full.double() * w).sum((1, 2)), z[f'abs{TT}_{ref_name}_wsum'], 1e-3)
    t = T(z['abs10_gather_t'])[:, 0].long()
    a = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    close(a['Qt_x'][t], z['abs10_gather_Qt_x']); close(a['Qtb_e'][t], z['abs10_gather_Qtb_e'])


def test_container_mask_and_collapse():
    z = load('container_masks')
    X, E = O.mask_dense(T(z['in_X']), T(z['in_E']), T(z['node_mask']))
    close(X, z['out_X'], 0); close(E, z['out_E'], 0)
    Xi, Ei = O.collapse_dense(T(z['in_X']), T(z['in_E']), T(z['node_mask']))
    assert np.array_equal(Xi.numpy(), z['col_X']) and np.array_equal(Ei.numpy(), z['col_E'])


@pytest.mark.parametrize('tag', ['rag', 'full'])
@pytest.mark.parametrize('mode', ['product_and_sn', 'atom_mapping', 'sn', 'product'])
def test_free_masks(tag, mode):
    z = load('container_masks')
    X = F.one_hot(T(z[f'{tag}_X_idx']), KX).float()
    mx, me = O.free_masks(X, T(z[f'{tag}_node_mask']), T(z[f'{tag}_am']), mode, suno_idx=28)
    mX, mE = O.expand_masks(mx, me, KX, KE)
    assert np.array_equal(mX[..., 0].numpy(), z[f'{tag}_{mode}_mask_X'])
    assert np.array_equal(mE[..., 0].numpy(), z[f'{tag}_{mode}_mask_E'])


def test_apply_mask_values():
    z = load('container_masks')
    X = F.one_hot(T(z['rag_X_idx']), KX).float(); E = F.one_hot(T(z['rag_E_idx']), KE).float()
    nm = T(z['rag_node_mask'])
    # the synthetic generator's dense encoding: padded rows / pairs / diagonal are class 0 one-hots
    mx, me = O.free_masks(X, nm, T(z['rag_am']), 'product_and_sn', suno_idx=28)
    for al in (0, 1):
        rX, rE = O.apply_mask(X, E, T(z['apply_zX']), T(z['apply_zE']), mx, me, as_logits=bool(al))
        close(rX, z[f'apply_{al}_X'], 0); close(rE, z[f'apply_{al}_E'], 0)


def test_layer_tiny():
    z = load('layer_tiny')
    sd = weights_of(z)
    X, E, y, nm = T(z['X']), T(z['E']), T(z['y']), T(z['node_mask'])
    H = int(z['n_head'])
    nX, nE, ny = O.node_edge_block(sd, 'self_attn', X, E, y, nm, H)
    close(nX, z['blk_X'], 2e-5); close(nE, z['blk_E'], 2e-5); close(ny, z['blk_y'], 2e-5)
    sd2 = {'L.' + k: v for k, v in sd.items()}
    oX, oE, oy = O.xey_layer(sd2, 'L', X, E, y, nm, H)
    close(oX, z['out_X'], 2e-5); close(oE, z['out_E'], 2e-5); close(oy, z['out_y'], 2e-5)


def _model_inputs(z):
    X = torch.cat([F.one_hot(T(z['X_idx']), KX).float(), T(z['X_extra'])], -1)
    E = F.one_hot(T(z['E_idx']), KE).float()
    return X, E, T(z['y_in']), T(z['node_mask']), T(z['am']), T(z['pos']), T(z['mol'])


def test_model_tiny_all_variants():
    z = load('model_tiny')
    sd = weights_of(z)
    X, E, y, nm, am, pos, mol = _model_inputs(z)
    L, H = int(z['n_layers']), int(z['n_head'])
    # plain / no-skip variants were generated with the same seed -> same weights minus skip params
    r = O.denoiser_forward(sd, X, E, y, nm, L, H, (KX, KE, 1))
    close(r[0], z['plain_X'], 5e-5); close(r[1], z['plain_E'], 5e-5); close(r[2], z['plain_y'], 5e-5)
    r = O.denoiser_forward(sd, X, E, y, nm, L, H, (KX, KE, 1), am, pos, mol, False)
    close(r[0], z['am_noskip_X'], 5e-5); close(r[1], z['am_noskip_E'], 5e-5)
    r = O.denoiser_forward(sd, X, E, y, nm, L, H, (KX, KE, 1), am, pos, mol, True)
    close(r[0], z['am_skip_X'], 5e-5); close(r[1], z['am_skip_E'], 5e-5); close(r[2], z['am_skip_y'], 5e-5)


def inalign_weights(z):
    """Weights of tests/golden/model_inalign.npz: the seeded initialiser on the recorded shapes (skip scalings 0.7 / 1.3)."""
    import ast
    from tests.golden.common import init_state_dict
    sd = init_state_dict({str(k): ast.literal_eval(str(s)) for k, s in zip(z['keys'], z['shapes'])}, seed=45)
    sd['skip_scaling'] = torch.tensor([0.7]); sd['skip_scaling_2'] = torch.tensor([1.3])
    return sd


def test_model_input_alignment():
    """GraphTransformerWithYAtomMapPosEmb(input_alignment=True), with and without the skip connection (:575-580, 630-647, 702-704)."""
    z = load('model_inalign')
    sd = inalign_weights(z)
    assert sd['mlp_in_X.0.weight'].shape[1] == 2 * KX + 8 and sd['mlp_in_E.0.weight'].shape[1] == 2 * KE
    X, E, y, nm, am, pos, mol = _model_inputs(z)
    L, H = int(z['n_layers']), int(z['n_head'])
    for tag, skip in (('noskip', False), ('skip', True)):
        r = O.denoiser_forward(sd, X, E, y, nm, L, H, (KX, KE, 1), am, pos, mol, skip, input_alignment=True)
        close(r[0], z[f'{tag}_X'], 5e-5); close(r[1], z[f'{tag}_E'], 5e-5); close(r[2], z[f'{tag}_y'], 5e-5)


def plain_weights_seed45():
    """Weights of the 'old_skip' case of model_inalign.npz (no input alignment): the seeded initialiser on the module's own shapes."""
    import torch.nn as nn
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
    from tests.golden.common import init_state_dict, tiny_dims
    d = tiny_dims()
    m = GraphTransformerWithYAtomMapPosEmb(d['n_layers'], {'X': KX + 8, 'E': KE, 'y': 13}, d['hidden_mlp_dims'], d['hidden_dims'],
                                           {'X': KX, 'E': KE, 'y': 1}, nn.ReLU(), nn.ReLU(), p_to_r_skip_connection=True, alignment_type='old')
    sd = init_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=45)
    sd['skip_scaling'] = torch.tensor([0.7]); sd['skip_scaling_2'] = torch.tensor([1.3])
    return sd


def test_model_alignment_type_old():
    """alignment_type='old' (:526-538): rank alignment, with the skip connection alone and with input_alignment."""
    z = load('model_inalign')
    X, E, y, nm, am, pos, mol = _model_inputs(z)
    L, H = int(z['n_layers']), int(z['n_head'])
    assert not torch.equal(O.align_index(am, mol), O.align_index_old(am, mol))      # the fixture tells the two alignments apart
    r = O.denoiser_forward(plain_weights_seed45(), X, E, y, nm, L, H, (KX, KE, 1), am, pos, mol, True, alignment_type='old')
    close(r[0], z['old_skip_X'], 5e-5); close(r[1], z['old_skip_E'], 5e-5)
    r = O.denoiser_forward(inalign_weights(z), X, E, y, nm, L, H, (KX, KE, 1), am, pos, mol, True, input_alignment=True, alignment_type='old')
    close(r[0], z['old_inal_skip_X'], 5e-5); close(r[1], z['old_inal_skip_E'], 5e-5)


def test_model_improved_block():
    """cfg.neuralnet.improved=True: NodeEdgeBlockImproved (layers.py:49-164) inside the atom-map model with the skip connection."""
    z = load('model_inalign')
    X, E, y, nm, am, pos, mol = _model_inputs(z)
    r = O.denoiser_forward(plain_weights_seed45(), X, E, y, nm, int(z['n_layers']), int(z['n_head']), (KX, KE, 1), am, pos, mol, True,
                           improved=True)
    close(r[0], z['improved_skip_X'], 5e-5); close(r[1], z['improved_skip_E'], 5e-5); close(r[2], z['improved_skip_y'], 5e-5)


def test_model_full_dims_one_layer():
    from tests.golden.common import init_state_dict
    import ast
    z = load('model_full1')
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(z['keys'], z['shapes'])}
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(z['n_params'])
    sd = init_state_dict(shapes, seed=54)
    X, E, y, nm, am, pos, mol = _model_inputs(z)
    r = O.denoiser_forward(sd, X, E, y, nm, 1, 8, (KX, KE, 1), am, pos, mol, True)
    close(r[0], z['out_X'], 2e-4); close(r[1], z['out_E'], 2e-4); close(r[2], z['out_y'], 2e-4)


def _tables(fam, z):
    if fam == 'abs':
        return O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    g = torch.Generator().manual_seed(11)
    xm = torch.rand(KX, generator=g); xm = xm / xm.sum()
    em = torch.rand(KE, generator=g); em = em / em.sum()
    return O.marginal_tables(10, xm, em, 1, NM, EM)


@pytest.mark.parametrize('fam', ['abs', 'mar'])
@pytest.mark.parametrize('t', [10, 5, 1])
def test_posterior_and_draws(fam, t):
    z = load('posterior_draws')
    tabs = _tables(fam, z)
    k = f'{fam}_t{t}_'
    nm = T(z[f'{fam}_node_mask'])
    B = nm.shape[0]
    ztX = F.one_hot(T(z[k + 'zt_X']), KX).float(); ztE = F.one_hot(T(z[k + 'zt_E']), KE).float()
    lX = torch.log_softmax(T(z[k + 'logit_X']), -1); lE = torch.log_softmax(T(z[k + 'logit_E']), -1)
    ex = lambda name, i: tabs[name][i][None].expand(B, -1, -1)
    pX, pE = O.posterior_probs(lX, lE, ztX, ztE, nm, ex('Qt_x', t), ex('Qtb_x', t - 1), ex('Qtb_x', t),
                               ex('Qt_e', t), ex('Qtb_e', t - 1), ex('Qtb_e', t))
    close(pX, z[k + 'prob_X'], 2e-6); close(pE, z[k + 'prob_E'], 2e-6)
    # draws: bit-exact given the reference's probabilities and the same noise
    c = int(z[k + 'noise_counter'])
    n = nm.shape[1]
    qX = counter_noise((B * n, KX), c).reshape(B, n, KX)
    qE = counter_noise((B * n * n, KE), c + 1).reshape(B, n, n, KE)
    Xs, Es = O.sample_discrete(T(z[k + 'prob_X']), T(z[k + 'prob_E']), nm, qX, qE)
    assert np.array_equal(Xs.numpy(), z[k + 'zs_X']) and np.array_equal(Es.numpy(), z[k + 'zs_E'])


def _tiny_setup():
    from diffalign_amd import synthetic as S
    z = load('chain_elbo_tiny')
    sd = weights_of(z)
    true = dict(X=F.one_hot(T(z['true_X']), KX).float(), E=F.one_hot(T(z['true_E']), KE).float(), node_mask=T(z['node_mask']),
                atom_map_numbers=T(z['am']), mol_assignments=T(z['mol']))
    pos = S.shared_pos_encodings(dict(atom_map_numbers=true['atom_map_numbers']), 64, seed=73)
    L_, H = int(z['n_layers']), int(z['n_head'])

    def fwd(zX, zE, y):
        r = O.denoiser_forward(sd, zX, zE, y, true['node_mask'], L_, H, (KX, KE, 1), true['atom_map_numbers'], pos,
                               true['mol_assignments'], True)
        return r[0], r[1]
    return z, sd, true, fwd


def test_single_step_logits_and_chain_config0():
    """BASELINE.json configs[0]: tiny 2-layer model, 10-step absorbing reverse diffusion on 4 synthetic graphs."""
    z, sd, true, fwd = _tiny_setup()
    lX, lE = fwd(F.one_hot(T(z['step_zt_X']), KX).float(), F.one_hot(T(z['step_zt_E']), KE).float(), torch.full((4, 1), 6.0))
    close(lX, z['step_logit_X'], 1e-4); close(lE, z['step_logit_E'], 1e-4)
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    zX, zE = O.sample_chain(fwd, true, tabs, 10, 10, 'product_and_sn', 28, lambda c, shp: counter_noise(shp, c))
    assert int(z['chain_noise_calls']) == 22
    assert np.array_equal(zX.argmax(-1).numpy(), z['chain_X'])
    assert np.array_equal(zE.argmax(-1).numpy(), z['chain_E'])


def test_temperature_chain_vs_reference():
    """tests/golden/chain_temp_tiny.npz (make_golden_r3.py): the reference's sample_one_batch(get_chains=True) at
    temperature_scaling_node = 0.7, temperature_scaling_edge = 1.4, log_every_t = 3 on configs[0]'s tiny model.  The oracle's chain
    equals it bit for bit, and at every logged step so do the state, p(z_s | z_t) after apply_mask and the prediction turned into
    probabilities the reference's way, softmax(temperature * log_softmax(logits)) without the supernode class (sampling.py:82,118-144)."""
    z0, sd, true, fwd = _tiny_setup()
    z = load('chain_temp_tiny')
    tx, te = (float(v) for v in z['temps'])
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    nz = lambda c, shp: counter_noise(shp, c)
    zX, zE = O.sample_chain(fwd, true, tabs, 10, 10, 'product_and_sn', 28, nz, temp=(tx, te))
    assert int(z['noise_calls']) == 22
    assert np.array_equal(zX.argmax(-1).numpy(), z['chain_X']) and np.array_equal(zE.argmax(-1).numpy(), z['chain_E'])
    assert not np.array_equal(z['chain_E'], z0['chain_E'])          # the temperatures do change the chain (configs[0] at 1.0)
    # the same loop step by step, with what the reference records at the logged steps
    X0, E0, nm = true['X'], true['E'], true['node_mask']
    B, n = nm.shape
    mx, me = O.free_masks(X0, nm, true['atom_map_numbers'], 'product_and_sn', 28)
    c = 0
    zX, zE = O.sample_from_limit(tabs['x_limit'], tabs['e_limit'], nm, nz(c, (B * n, KX)).reshape(B, n, KX), nz(c + 1, (B * n * n, KE)).reshape(B, n, n, KE))
    zX, zE = O.apply_mask(X0, E0, zX, zE, mx, me)
    c += 2
    logged = [int(s) for s in z['logged_steps']]
    assert logged == [9, 6, 3, 0] and [int(t) for t in z['sample_steps']] == [10, 7, 4, 1, 0]
    for t in range(10, 0, -1):
        lX, lE = fwd(zX, zE, torch.full((B, 1), float(t)))
        qX, qE = nz(c, (B * n, KX)).reshape(B, n, KX), nz(c + 1, (B * n * n, KE)).reshape(B, n, n, KE)
        c += 2
        pX, pE, nX, nE = O.reverse_step(lX, lE, zX, zE, nm, tabs, t, X0, E0, mx, me, qX, qE, tx, te)
        s = t - 1
        if s in logged:
            cX, cE = O.collapse_dense(zX, zE, nm)
            assert np.array_equal(np.asarray(cX), z[f'sample{t}_X']) and np.array_equal(np.asarray(cE), z[f'sample{t}_E'])
            mpX, mpE = O.apply_mask(X0, E0, pX, pE, mx, me)
            close(mpX, z[f'prob{s}_X'], 2e-6); close(mpE, z[f'prob{s}_E'], 2e-6)
            p0X = torch.softmax(tx * torch.log_softmax(lX, -1), -1)
            p0E = torch.softmax(te * torch.log_softmax(lE, -1), -1)
            p0X[..., -1] = 0.
            p0X = p0X / p0X.sum(-1, keepdim=True)
            m0X, m0E = O.apply_mask(X0, E0, p0X, p0E, mx, me)
            close(m0X, z[f'pred{s}_X'], 2e-6); close(m0E, z[f'pred{s}_E'], 2e-6)
        zX, zE = nX, nE


def oracle_training_loss(z, sd, true, kind, pos=None, requires_grad=True):
    """The three training objectives (diffusion_abstract.py:198-222) on the reference's recorded (t, z_t), differentiable in `sd`."""
    from diffalign_amd import synthetic as S
    if pos is None:
        pos = S.shared_pos_encodings(dict(atom_map_numbers=true['atom_map_numbers']), 64, seed=73)
    t = T(z['train_t']).float()
    ztX, ztE = F.one_hot(T(z['train_zt_X']), KX).float(), F.one_hot(T(z['train_zt_E']), KE).float()
    lX, lE, _ = O.denoiser_forward(sd, ztX, ztE, t, true['node_mask'], int(z['n_layers']), int(z['n_head']), (KX, KE, 1),
                                   true['atom_map_numbers'], pos, true['mol_assignments'], True)
    mx, me = O.free_masks(true['X'], true['node_mask'], true['atom_map_numbers'], 'product_and_sn', 28)
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    ce = O.ce_loss(lX, lE, true['X'], true['E'], mx, me, 5.0)[2]
    if kind == 'ce':
        return ce, (lX, lE)
    vb = O.elbo_batch_quick(true['X'], true['E'], ztX, ztE, lX, lE, tabs, t[:, 0], 10, mx, me, 5.0)
    return (vb if kind == 'vb' else vb + 2 * ce), (lX, lE)


GNORM_PARAMS = ('mlp_in_E.0.weight', 'tf_layers.0.self_attn.e_out.weight', 'tf_layers.1.linE2.weight', 'mlp_out_X.2.weight', 'skip_scaling')


@pytest.mark.parametrize('kind', ['ce', 'vb', 'vbce'])
def test_training_losses_and_gradient_norms(kind):
    """autograd through the oracle reproduces the loss values and the parameter-gradient norms that the reference's
    own `loss.backward()` left (make_golden.py:432-469) -- this pins the checker the HIP backward is compared with."""
    z, sd, true, _ = _tiny_setup()
    sd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    loss, _ = oracle_training_loss(z, sd, true, kind)
    loss.backward()
    assert abs(float(loss) - float(z[f'train_{kind}_loss'])) <= 2e-4 * abs(float(z[f'train_{kind}_loss'])) + 1e-4
    for pn in GNORM_PARAMS:
        want, got = float(z[f'train_{kind}_gnorm::{pn}']), float(sd[pn].grad.norm())
        assert abs(got - want) <= 1e-3 * abs(want) + 1e-6, (pn, got, want)
    for pn in ('mlp_out_y.0.weight', 'tf_layers.0.self_attn.y_y.weight', 'skip_scaling_3'):
        assert sd[pn].grad is None


def test_graph_utils_fix_duplicate_to_dense():
    """utils/graph.py fix_nodes_and_edges_by_idx / duplicate_data / to_dense against the reference's outputs."""
    z = load('graph_utils')
    mX, mE = O.fix_masks(3, 24, [[0, 1, 5], [2], []], [[[0, 1], [1, 5]], [[2, 3]], []])
    assert np.array_equal(mX.numpy(), z['fix_mask_X']) and np.array_equal(mE.numpy(), z['fix_mask_E'])
    d = dict(X=T(z['dup_in_X']), E=F.one_hot(T(z['fix_E_idx']), KE).float(), node_mask=T(z['fix_node_mask']), y=torch.zeros(3, 1))
    dd = O.duplicate(d, 3)
    close(dd['X'], z['dup_X'], 0); assert np.array_equal(dd['E'].argmax(-1).numpy(), z['dup_E_idx'])
    assert np.array_equal(dd['node_mask'].numpy(), z['dup_node_mask'])
    cX, cE = O.collapse_dense(dd['X'], dd['E'], dd['node_mask'])
    assert np.array_equal(cX.numpy(), z['dup_disc_X']) and np.array_equal(cE.numpy(), z['dup_disc_E'])
    td = O.to_dense(F.one_hot(T(z['td_x']), KX).float(), T(z['td_batch']), T(z['td_edge_index']), F.one_hot(T(z['td_edge_attr']), KE).float(),
                    T(z['td_am']), T(z['td_mol']))
    close(td['X'], z['td_X'], 0); close(td['E'], z['td_E'], 0)
    assert np.array_equal(td['node_mask'].numpy(), z['td_node_mask'])
    assert np.array_equal(td['atom_map_numbers'].numpy(), z['td_am_dense']) and np.array_equal(td['mol_assignments'].numpy(), z['td_mol_dense'])


INPAINT_NODES = [[0, 1, 2], [3], [], [0, 4]]
INPAINT_EDGES = [[[0, 1], [1, 2]], [], [[1, 2]], [[0, 4]]]


def test_inpainting_chain_and_sample_for_condition():
    """sample_one_batch with inpainting lists and sample_for_condition (2 conditions x 3 samples): reproduced bit for bit."""
    from diffalign_amd import synthetic as S
    zi = load('inpaint_tiny')
    z, sd, true, fwd = _tiny_setup()
    assert np.array_equal(zi['true_X'], z['true_X'])
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    zX, zE = O.sample_chain(fwd, true, tabs, 10, 10, 'product_and_sn', 28, lambda c, shp: counter_noise(shp, c), node_idx=INPAINT_NODES,
                            edge_idx=INPAINT_EDGES)
    assert np.array_equal(zX.argmax(-1).numpy(), zi['inp_X']) and np.array_equal(zE.argmax(-1).numpy(), zi['inp_E'])
    for b, idx in enumerate(INPAINT_NODES):                               # the property inpainting exists for
        assert torch.equal(zX[b, idx], true['X'][b, idx])
    dup = O.duplicate({k: v[:2] for k, v in true.items()}, 3)
    pos = S.shared_pos_encodings(dict(atom_map_numbers=dup['atom_map_numbers']), 64, seed=73)

    def fwd6(zX, zE, y):
        r = O.denoiser_forward(sd, zX, zE, y, dup['node_mask'], int(z['n_layers']), int(z['n_head']), (KX, KE, 1), dup['atom_map_numbers'], pos,
                               dup['mol_assignments'], True)
        return r[0], r[1]
    rep = lambda idx: [i for i in idx for _ in range(3)]
    sX, sE = O.sample_chain(fwd6, dup, tabs, 10, 10, 'product_and_sn', 28, lambda c, shp: counter_noise(shp, 100 + c),
                            node_idx=rep([[0, 1], []]), edge_idx=rep([[[0, 1]], []]))
    cX, cE = O.collapse_dense(sX, sE, dup['node_mask'])
    assert np.array_equal(cX.numpy(), zi['sfc_X']) and np.array_equal(cE.numpy(), zi['sfc_E'])


def test_counter_noise_is_exponential_and_layout_free():
    seeds = torch.tensor([5, 123456789012, -7], dtype=torch.int64)
    qX, qE = O.counter_exp_noise(seeds, 3, 40, KX, KE)
    assert qX.shape == (3, 40, KX) and qE.shape == (3, 40, 40, KE) and (qE > 0).all() and torch.isfinite(qE).all()
    assert abs(qE.mean().item() - 1.0) < 0.02 and abs(qE.var().item() - 1.0) < 0.05          # Exp(1)
    a, _ = O.counter_exp_noise(seeds[1:2], 3, 40, KX, KE)
    assert torch.equal(a[0], qX[1])                                       # a chain's noise does not depend on its batch position
    b, _ = O.counter_exp_noise(seeds, 4, 40, KX, KE)
    assert not torch.equal(b, qX)


@pytest.mark.parametrize('case', ['rxn_true', 'rxn_t5', 'conn', 'dense', 'sparse'])
def test_extra_features(case):
    """a8: oracle restatement of ExtraFeatures('all') + ExtraMolecularFeatures vs the reference's outputs."""
    from types import SimpleNamespace
    from tests.golden.common import extra_info, extra_case, check_extra
    z = load('extra_features')
    info = extra_info(SimpleNamespace)
    X, E, nm, pad_noise, _, _ = extra_case(z, case)
    Xe, ye = O.extra_features(X, E, nm, info.max_n_nodes, info.valencies, info.bond_orders, info.atom_weights.values(),
                              info.max_weight, pad_noise)
    Lap = O.eigen_features(E, nm, pad_noise)[4]
    check_extra(Xe, ye, case, z, E, nm, Lap)
    # same LAPACK as the generator: the basis-dependent outputs agree too
    torch.testing.assert_close(Xe, torch.from_numpy(z[f'{case}::X_extra']), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_pos_encodings(tag):
    """a13: sinusoidal product-index, info-leak and Gaussian positional encodings vs the reference's PositionalEmbedding
    ('b' has unmapped and duplicated product atom-map numbers).  sin/cos of arguments up to 5e4: 1e-6 absolute."""
    z = load('pos_encodings')
    dim = int(z['dim'])
    am, mol = torch.from_numpy(z[f'{tag}::am']), torch.from_numpy(z[f'{tag}::mol'])
    torch.testing.assert_close(O.pos_enc_sinusoidal(am, mol, dim), torch.from_numpy(z[f'{tag}::sinusoidal']), atol=1e-6, rtol=0)
    torch.testing.assert_close(O.pos_enc_infoleak(am, dim), torch.from_numpy(z[f'{tag}::infoleak']), atol=1e-6, rtol=0)
    assert torch.equal(O.pos_enc_gaussian(am, torch.from_numpy(z[f'{tag}::gaussian_noise'])), torch.from_numpy(z[f'{tag}::gaussian']))


@pytest.mark.parametrize('variant', ['laplacian', 'laplacian_gpu'])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_laplacian_pos_encodings_vs_reference(tag, variant):
    """a13: oracle restatement against the reference's two Laplacian variants (scipy eigsh and torch eigh), generated by
    tests/golden/make_golden_r2.py on shims for the three torch_geometric helpers (see its docstring)."""
    from tests.golden.common import check_lap_encodings
    z = load('pos_laplacian')
    g = lambda k: torch.from_numpy(z[f'{tag}::{k}'])
    dim, K = int(z['dim']), int(z['K'])
    out, laps = O.pos_enc_laplacian(g('E'), g('am'), g('mol'), K, dim, g(f'{variant}_signs'))
    torch.testing.assert_close(laps, g(f'{variant}_lap'), atol=1e-6, rtol=0)       # the product-block Laplacian itself
    done, total = check_lap_encodings(out, g(variant), laps, g('am'), g('mol'), K)
    assert done == total and total >= 15, (done, total)     # every well-defined column was compared


CFG_NODES, CFG_EDGES = [[0, 1], [], [], [2]], [[[0, 1]], [], [], []]


def _uncond_forward(z, sd, true, full):
    """The denoiser as classifier-free guidance calls it without a condition: positional encodings off, and with
    classifier_free_full_unconditioning the atom-map numbers cleared."""
    am = torch.zeros_like(true['atom_map_numbers']) if full else true['atom_map_numbers']
    pos0 = torch.zeros(*true['node_mask'].shape, 64)

    def fwd(zX, zE, y):
        r = O.denoiser_forward(sd, zX, zE, y, true['node_mask'], int(z['n_layers']), int(z['n_head']), (KX, KE, 1), am, pos0,
                               true['mol_assignments'], True)
        return r[0], r[1]
    return fwd


@pytest.mark.parametrize('full', [False, True])
def test_classifier_free_guidance_chain(full):
    """sample_one_batch with classifier_free_guidance (sampling.py:88-113), weight 0.7, inpainting lists: bit for bit."""
    zc = load('cfg_tiny')
    z, sd, true, fwd = _tiny_setup()
    assert np.array_equal(zc['true_X'], z['true_X'])
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    g = dict(w=float(zc['weight']), forward_uncond=_uncond_forward(z, sd, true, full), tabs_train=tabs)
    zX, zE = O.sample_chain(fwd, true, tabs, 10, 10, 'product_and_sn', 28, lambda c, shp: counter_noise(shp, c), node_idx=CFG_NODES,
                            edge_idx=CFG_EDGES, guidance=g)
    assert np.array_equal(zX.argmax(-1).numpy(), zc[f'chain_full{int(full)}_X'])
    assert np.array_equal(zE.argmax(-1).numpy(), zc[f'chain_full{int(full)}_E'])


def oracle_cfg_training(zc, z, sd, true, within, full):
    """training_step with classifier-free guidance (diffusion_abstract.py:152-185) restated on the oracle's pieces: returns
    (loss, z_t X, z_t E, use_pos, atom maps) for the reference's recorded t and r."""
    from diffalign_amd import synthetic as S
    t, r = T(zc['train_t']).float(), T(zc['train_r'])
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, NM, EM)
    nm = true['node_mask']
    B, n = nm.shape
    pX, pE = O.noise_probs(true['X'], true['E'], nm, tabs['Qtb_x'][t[:, 0].long()], tabs['Qtb_e'][t[:, 0].long()])
    Xs, Es = O.sample_discrete(pX, pE, nm, counter_noise((B * n, KX), 7000).reshape(B, n, KX),
                               counter_noise((B * n * n, KE), 7001).reshape(B, n, n, KE))
    mx, me = O.free_masks(true['X'], nm, true['atom_map_numbers'], 'product_and_sn', 28)
    ztX, ztE = O.apply_mask(true['X'], true['E'], *O.one_hot_pair(Xs, Es, KX, KE), mx, me)
    fx, fe = O.free_masks(true['X'], nm, true['atom_map_numbers'], 'product_and_sn', 28, include_supernode=False)
    uncond = r < 0.1
    keep = ~uncond if within else (~uncond[0:1]).expand(B)
    MX, ME = O.expand_masks(fx | keep[:, None], fe | keep[:, None], KX, KE)
    am = true['atom_map_numbers'] * uncond[:, None].long() if full else true['atom_map_numbers']
    pos = S.shared_pos_encodings(dict(atom_map_numbers=am), 64, seed=73) * (~uncond)[:, None, None].float()
    inX, inE = ztX * MX, ztE * ME
    lX, lE, _ = O.denoiser_forward(sd, inX, inE, t, nm, int(z['n_layers']), int(z['n_head']), (KX, KE, 1), am, pos, true['mol_assignments'], True)
    return O.ce_loss(lX, lE, true['X'], true['E'], mx, me, 5.0)[2], inX, inE, ~uncond, am


@pytest.mark.parametrize('within,full', [(True, False), (True, True), (False, False), (False, True)])
def test_classifier_free_guidance_training_step(within, full):
    zc = load('cfg_tiny')
    z, sd, true, _ = _tiny_setup()
    sd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    loss, inX, inE, use, am = oracle_cfg_training(zc, z, sd, true, within, full)
    k = f'train_w{int(within)}_f{int(full)}_'
    close(inX, zc[k + 'zt_X'], 0); close(inE, zc[k + 'zt_E'], 0)
    assert np.array_equal(use.numpy(), zc[k + 'use']) and np.array_equal((am == 0).numpy(), zc[k + 'am_zero'])
    loss.backward()
    assert abs(float(loss.detach()) - float(zc[k + 'loss'])) <= 2e-4 * abs(float(zc[k + 'loss']))
    for pn in ('mlp_in_E.0.weight', 'tf_layers.1.linE2.weight', 'mlp_out_X.2.weight', 'skip_scaling'):
        want, got = float(zc[k + 'gnorm::' + pn]), float(sd[pn].grad.norm())
        assert abs(got - want) <= 1e-3 * abs(want) + 1e-6, (pn, got, want)
