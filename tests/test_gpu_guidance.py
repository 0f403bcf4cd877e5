"""GPU: classifier-free guidance (diffusion/sampling.py:88-113, diffusion_abstract.py:152-185) against the
reference's own sample_one_batch and training_step (tests/golden/cfg_tiny.npz, tests/golden/make_golden_r2.py gen_cfg).

Tolerances: the sampling chain runs the bf16 inference path twice per step (conditional + unconditional forward), so near-tie
draws may flip and agreement with the reference's final graphs is statistical (the fp32 oracle reproduces them bit for bit:
tests/test_oracle_golden.py); the training step is fp32: loss 2e-5 relative, gradient norms 5e-4 relative."""
import numpy as np
import pytest
import torch

from tests.golden.common import load, counter_noise
from tests.test_gpu_elbo import setup, T, KX, KE, D

pytestmark = pytest.mark.gpu
CFG_NODES, CFG_EDGES = [[0, 1], [], [], [2]], [[[0, 1]], [], [], []]


def guided_model(**kw):
    z, model, data, pos = setup()
    d = model.cfg.diffusion
    d.classifier_free_guidance, d.classifier_free_guidance_weight = True, 0.7
    d.classifier_free_full_unconditioning, d.classifier_free_uncond_prob, d.classifier_free_drop_within_batch = False, 0.1, False
    for k, v in kw.items():
        setattr(d, k, v)
    return z, model, data, pos


@pytest.mark.parametrize('full', [False, True])
def test_guided_chain_against_reference_golden(full):
    zc = load('cfg_tiny')
    z, model, data, _ = guided_model(classifier_free_full_unconditioning=full)
    assert np.array_equal(zc['true_X'], z['true_X'])
    out = model.sample_one_batch(data=data, noise_fn=lambda c, shp: counter_noise(shp, c), inpaint_node_idx=CFG_NODES,
                                 inpaint_edge_idx=CFG_EDGES)
    gx, ge = out.X.argmax(-1).cpu().numpy(), out.E.argmax(-1).cpu().numpy()
    ax, ae = (gx == zc[f'chain_full{int(full)}_X']).mean(), (ge == zc[f'chain_full{int(full)}_E']).mean()
    print(f'guided chain (full unconditioning {full}) vs reference: nodes {ax:.4f}, edges {ae:.4f}')
    assert ax >= 0.98 and ae >= 0.99, (ax, ae)
    assert torch.equal(out.E, out.E.transpose(1, 2)) and bool((out.X.sum(-1) == 1).all()) and bool((out.E.sum(-1) == 1).all())
    for b, idx in enumerate(CFG_NODES):
        assert torch.equal(out.X[b, idx], data.X[b, idx])
    # guidance changes the draw: the unguided chain on the same noise ends elsewhere
    model.cfg.diffusion.classifier_free_guidance = False
    plain = model.sample_one_batch(data=data, noise_fn=lambda c, shp: counter_noise(shp, c), inpaint_node_idx=CFG_NODES,
                                   inpaint_edge_idx=CFG_EDGES)
    assert not torch.equal(plain.E, out.E)


@pytest.mark.parametrize('within,full', [(True, False), (True, True), (False, False), (False, True)])
def test_guided_training_step_against_reference_golden(within, full):
    from diffalign_amd import synthetic as S
    zc = load('cfg_tiny')
    z, model, data, _ = guided_model(classifier_free_full_unconditioning=full, classifier_free_drop_within_batch=within)
    model.pos_encodings_fn = lambda zt: S.shared_pos_encodings(dict(atom_map_numbers=zt.atom_map_numbers.cpu()), 64, seed=73).to(D)
    model.train()
    B, n = data.node_mask.shape
    noise = (counter_noise((B * n, KX), 7000).reshape(B, n, KX).to(D), counter_noise((B * n * n, KE), 7001).reshape(B, n, n, KE).to(D))
    seen = {}
    fwd = model.forward

    def forward(z_t, use_pos_encoding_if_applicable=None, pos_encodings=None, step=None):
        seen.update(X=z_t.X.clone(), E=z_t.E.clone(), use=use_pos_encoding_if_applicable.clone(), am=z_t.atom_map_numbers.clone())
        return fwd(z_t=z_t, use_pos_encoding_if_applicable=use_pos_encoding_if_applicable, pos_encodings=pos_encodings)
    model.forward = forward
    _, _, loss = model.training_step(data, 1, t_int=T(zc['train_t']).to(D), noise=noise, uncond_draw=T(zc['train_r']))
    k = f'train_w{int(within)}_f{int(full)}_'
    assert np.array_equal(seen['X'].cpu().numpy(), zc[k + 'zt_X']) and np.array_equal(seen['E'].cpu().numpy(), zc[k + 'zt_E'])
    assert np.array_equal(seen['use'].cpu().numpy(), zc[k + 'use']) and np.array_equal((seen['am'] == 0).cpu().numpy(), zc[k + 'am_zero'])
    loss.backward()
    want = float(zc[k + 'loss'])
    assert abs(float(loss.detach()) - want) <= 2e-5 * abs(want) + 1e-5, (float(loss.detach()), want)
    params = dict(model.model.named_parameters())
    for pn in ('mlp_in_E.0.weight', 'tf_layers.1.linE2.weight', 'mlp_out_X.2.weight', 'skip_scaling'):
        wn, got = float(zc[k + 'gnorm::' + pn]), float(params[pn].grad.norm())
        assert abs(got - wn) <= 5e-4 * abs(wn) + 1e-6, (pn, got, wn)
