"""GPU: GraphTransformerWithYAtomMapPosEmb(input_alignment=True) (transformer_model_with_y.py:575-580, 630-647,
702-704) -- the aligned product features as extra input channels, with and without the product->reactant skip connection.
Inference (bf16 MFMA path) against the reference's logits (tests/golden/model_inalign.npz) and the oracle at full dims: rel-L2 2e-2
as for the other model tests; the fp32 training forward and its parameter gradients against autograd through the oracle: 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import load, init_state_dict
from tests.test_gpu_denoiser import TINY, FULL, IN_DIMS, OUT_DIMS, KX, KE, T, D, rel_l2, inputs
from tests.test_oracle_golden import inalign_weights

pytestmark = pytest.mark.gpu


def build(dims, n_layers, skip, sd=None, seed=45, align='correct', inal=True, improved=False):
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
    m = GraphTransformerWithYAtomMapPosEmb(n_layers, IN_DIMS, dims['hidden_mlp_dims'], dims['hidden_dims'], OUT_DIMS, nn.ReLU(), nn.ReLU(),
                                           dropout=0.0, p_to_r_skip_connection=skip, p_to_r_init=1.0, alignment_type=align,
                                           input_alignment=inal, improved=improved).eval()
    if sd is None:
        sd = init_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
        if skip:
            sd['skip_scaling'] = torch.tensor([0.7]); sd['skip_scaling_2'] = torch.tensor([1.3])
    m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()})
    return m.to(D), sd


@pytest.mark.parametrize('skip', [False, True])
def test_inference_vs_reference_golden(skip):
    z = load('model_inalign')
    m, _ = build(TINY, int(z['n_layers']), skip, sd=inalign_weights(z))
    assert m.mlp_in_X[0].in_features == 2 * KX + 8 and m.mlp_in_E[0].in_features == 2 * KE
    X = torch.cat([F.one_hot(T(z['X_idx']), KX).float(), T(z['X_extra'])], -1)
    E = F.one_hot(T(z['E_idx']), KE).float()
    with torch.no_grad():
        r = m(X.to(D), E.to(D), T(z['y_in']).to(D), T(z['node_mask']).to(D), T(z['am']).to(D), T(z['pos']).to(D), T(z['mol']).to(D))
    tag = 'skip' if skip else 'noskip'
    for a_, name in ((r[0], 'X'), (r[1], 'E'), (r[2], 'y')):
        assert rel_l2(a_, T(z[f'{tag}_{name}'])) <= 2e-2, (name, rel_l2(a_, T(z[f'{tag}_{name}'])))
    assert torch.equal(r[1], r[1].transpose(1, 2))


def test_full_dims_vs_oracle_and_fused_step():
    from diffalign_amd import helpers
    from diffalign_amd.placeholder import PlaceHolder
    m, sd = build(FULL, 2, True, seed=46)
    X, E, y, nm, am, pos, mol = inputs(2, 70, seed=33, dx=256)
    dv = lambda t: t.to(D)
    with torch.no_grad():
        r = m(dv(X), dv(E), dv(y), dv(nm), dv(am), dv(pos), dv(mol))
    o = O.denoiser_forward(sd, X, E, y, nm, 2, 8, (KX, KE, 1), am, pos, mol, True, input_alignment=True)
    for a_, b_, name in ((r[0], o[0], 'X'), (r[1], o[1], 'E'), (r[2], o[2], 'y')):
        assert rel_l2(a_, b_) <= 2e-2, (name, rel_l2(a_, b_))
    # the reverse step fused behind the output stage draws what the separate posterior kernel draws from these logits
    B, n = nm.shape
    zt = PlaceHolder(X=dv(X[..., :KX]), E=dv(E), y=dv(y), node_mask=dv(nm), atom_map_numbers=dv(am), mol_assignments=dv(mol))
    g = torch.Generator().manual_seed(5)
    Q = lambda k: torch.softmax(torch.randn(B, k, k, generator=g), -1).to(D)
    Qs = [PlaceHolder(X=Q(KX), E=Q(KE), y=None) for _ in range(3)]
    noise = (torch.empty(B, n, KX).exponential_(generator=g).to(D), torch.empty(B, n, n, KE).exponential_(generator=g).to(D))
    ctx = helpers.reverse_step_ctx(zt, *Qs, noise=noise)
    with torch.no_grad():
        m(dv(X), dv(E), dv(y), dv(nm), dv(am), dv(pos), dv(mol), step=ctx)
    _, _, zsX, zsE = helpers.posterior_sample(r[0], r[1], zt, *Qs, noise=noise)
    assert torch.equal(ctx['zs_X'], zsX) and torch.equal(ctx['zs_E'], zsE)


@pytest.mark.parametrize('skip', [False, True])
def test_training_forward_and_gradients_vs_oracle(skip):
    m, sd = build(TINY, 2, skip, seed=47)
    m.train()
    X, E, y, nm, am, pos, mol = inputs(3, 24, seed=35, dx=64)
    dv = lambda t: t.to(D)
    r = m(dv(X), dv(E), dv(y), dv(nm), dv(am), dv(pos), dv(mol))
    osd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o = O.denoiser_forward(osd, X, E, y, nm, 2, 4, (KX, KE, 1), am, pos, mol, skip, input_alignment=True)
    assert (r[0].detach().cpu() - o[0].detach()).abs().max() <= 1e-4 and (r[1].detach().cpu() - o[1].detach()).abs().max() <= 1e-4
    g = torch.Generator().manual_seed(6)
    wX, wE = torch.randn(o[0].shape, generator=g), torch.randn(o[1].shape, generator=g)
    ((o[0] * wX).sum() + (o[1] * wE).sum()).backward()
    ((r[0] * dv(wX)).sum() + (r[1] * dv(wE)).sum()).backward()
    for name, p in m.named_parameters():
        want = osd[name].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        assert err <= 1e-4, (name, err)


@pytest.mark.parametrize('inal', [False, True])
def test_alignment_type_old_vs_reference_golden_and_training(inal):
    """alignment_type='old' (:526-538, rank alignment): inference against the reference's logits, the training forward and its
    gradients against the oracle."""
    from tests.test_oracle_golden import plain_weights_seed45
    z = load('model_inalign')
    sd = inalign_weights(z) if inal else plain_weights_seed45()
    m, _ = build(TINY, int(z['n_layers']), True, sd=sd, align='old', inal=inal)
    X = torch.cat([F.one_hot(T(z['X_idx']), KX).float(), T(z['X_extra'])], -1)
    E = F.one_hot(T(z['E_idx']), KE).float()
    args = (T(z['y_in']), T(z['node_mask']), T(z['am']), T(z['pos']), T(z['mol']))
    with torch.no_grad():
        r = m(X.to(D), E.to(D), *[a.to(D) for a in args])
    tag = 'old_inal_skip' if inal else 'old_skip'
    for a_, name in ((r[0], 'X'), (r[1], 'E')):
        assert rel_l2(a_, T(z[f'{tag}_{name}'])) <= 2e-2, (name, rel_l2(a_, T(z[f'{tag}_{name}'])))
    m.train()
    rt = m(X.to(D), E.to(D), *[a.to(D) for a in args])
    assert (rt[0].detach().cpu() - T(z[f'{tag}_X'])).abs().max() <= 2e-4 and (rt[1].detach().cpu() - T(z[f'{tag}_E'])).abs().max() <= 2e-4
    osd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o = O.denoiser_forward(osd, X, E, args[0], args[1], int(z['n_layers']), int(z['n_head']), (KX, KE, 1), args[2], args[3], args[4], True,
                           input_alignment=inal, alignment_type='old')
    (o[0].sum() + (o[1] ** 2).sum()).backward()
    (rt[0].sum() + (rt[1] ** 2).sum()).backward()
    for name in ('skip_scaling', 'skip_scaling_2', 'mlp_in_E.0.weight', 'mlp_out_E.2.weight'):
        want, got = osd[name].grad, dict(m.named_parameters())[name].grad.cpu()
        assert float((got - want).norm()) <= 1e-4 * float(want.norm()) + 1e-6, name


def test_improved_block_vs_reference_golden_and_training():
    """cfg.neuralnet.improved=True (NodeEdgeBlockImproved, layers.py:49-164): no fused kernels -- eval and train both run the fp32 path
    of train_forward.py (HIP autograd.Functions + library GEMMs).  Logits against the reference's to 2e-4, every parameter gradient
    against autograd through the oracle to 1e-4; y_e_mul / y_e_add exist in the block but are unused and get no gradient."""
    from tests.test_oracle_golden import plain_weights_seed45
    z = load('model_inalign')
    sd = plain_weights_seed45()
    m, _ = build(TINY, int(z['n_layers']), True, sd=sd, inal=False, improved=True)
    assert not m.supports_fused_step()
    X = torch.cat([F.one_hot(T(z['X_idx']), KX).float(), T(z['X_extra'])], -1)
    E = F.one_hot(T(z['E_idx']), KE).float()
    args = (T(z['y_in']), T(z['node_mask']), T(z['am']), T(z['pos']), T(z['mol']))
    with torch.no_grad():
        r = m(X.to(D), E.to(D), *[a.to(D) for a in args])
    for a_, name in ((r[0], 'X'), (r[1], 'E'), (r[2], 'y')):
        assert (a_.cpu() - T(z[f'improved_skip_{name}'])).abs().max() <= 2e-4, name
    m.train()
    rt = m(X.to(D), E.to(D), *[a.to(D) for a in args])
    osd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o = O.denoiser_forward(osd, X, E, args[0], args[1], int(z['n_layers']), int(z['n_head']), (KX, KE, 1), args[2], args[3], args[4], True,
                           improved=True)
    g = torch.Generator().manual_seed(8)
    wX, wE = torch.randn(o[0].shape, generator=g), torch.randn(o[1].shape, generator=g)
    ((o[0] * wX).sum() + (o[1] * wE).sum()).backward()
    ((rt[0] * wX.to(D)).sum() + (rt[1] * wE.to(D)).sum()).backward()
    for name, p in m.named_parameters():
        want = osd[name].grad
        if want is None or float(want.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        assert err <= 1e-4, (name, err)
    assert dict(m.named_parameters())['tf_layers.0.self_attn.y_e_mul.weight'].grad is None
