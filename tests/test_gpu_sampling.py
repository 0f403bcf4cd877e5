"""GPU: the reverse-diffusion loop (SamplingMixin.sample_one_batch mirror) against the reference's golden chain
(BASELINE.json configs[0]) and size-independent properties at MAX_ATOMS_RXN."""
from types import SimpleNamespace as ns

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden.common import load, weights_of, counter_noise, tiny_dims, ATOMS, BONDS

pytestmark = pytest.mark.gpu
KX, KE = len(ATOMS), len(BONDS)
T = lambda a: torch.from_numpy(np.asarray(a))
D = 'cuda:0'


def make_cfg(dims, T_=10, T_eval=10, extra=False):
    return ns(diffusion=ns(diffusion_steps=T_, diffusion_steps_eval=T_eval, transition='absorbing_masknoedge',
                           mask_nodes='product_and_sn', mask_edges='product_and_sn', node_states_to_mask=['SuNo', 'none'],
                           edge_states_to_mask=['mol', 'within', 'across'], diffuse_nodes=True, diffuse_edges=True,
                           temperature_scaling_node=1.0, temperature_scaling_edge=1.0, classifier_free_guidance=False,
                           lambda_train=[5, 0], lambda_test=1, ce_lambda=2, denoiser='neuralnet',
                           diffusion_noise_schedule='cosine'),
              neuralnet=ns(architecture='with_y_atommap_number_pos_enc', n_layers=dims['n_layers'], hidden_dims=dims['hidden_dims'],
                           hidden_mlp_dims=dims['hidden_mlp_dims'], dropout=0.0, improved=False, extra_features=extra,
                           pos_encoding_type='laplacian_pos_enc', p_to_r_skip_connection=True, p_to_r_init=1.0,
                           alignment_type='correct', input_alignment=False, increase_y_dim_for_multigpu=True, use_ema=False,
                           pos_emb_permutations=0),
              train=ns(loss='ce', log_every_t=1), test=ns(with_denoising=True, loss_0_repeat=1))


def make_info():
    return ns(atom_decoder=list(ATOMS), bond_decoder=list(BONDS), input_dims={'X': KX, 'E': KE, 'y': 1},
              output_dims={'X': KX, 'E': KE, 'y': 0}, nodes_dist=None, node_types=torch.ones(KX), edge_types=torch.ones(KE))


def test_chain_config0_against_reference_golden():
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    z = load('chain_elbo_tiny')
    model = DiscreteDenoisingDiffusion(make_cfg(tiny_dims()), make_info(), device=D).eval()
    model.model.load_state_dict(weights_of(z))
    am = T(z['am'])
    pos = S.shared_pos_encodings(dict(atom_map_numbers=am), 64, seed=73).to(D)
    model.pos_encodings_fn = lambda zt: pos
    data = PlaceHolder(X=F.one_hot(T(z['true_X']), KX).float().to(D), E=F.one_hot(T(z['true_E']), KE).float().to(D),
                       y=torch.zeros(4, 1, device=D), node_mask=T(z['node_mask']).to(D), atom_map_numbers=am.to(D),
                       mol_assignments=T(z['mol']).to(D))
    # one step, teacher-forced on the reference's z_t: logits within the bf16 tolerance
    zt = data.get_new_object(X=F.one_hot(T(z['step_zt_X']), KX).float().to(D), E=F.one_hot(T(z['step_zt_E']), KE).float().to(D),
                             y=torch.full((4, 1), 6.0, device=D))
    with torch.no_grad():
        pred = model.forward(zt, pos_encodings=pos)
    rel = lambda a, b: ((a.cpu().double() - b.double()).norm() / b.double().norm()).item()
    assert rel(pred.X, T(z['step_logit_X'])) <= 2e-2 and rel(pred.E, T(z['step_logit_E'])) <= 2e-2
    # the full 10-step chain with the reference's noise: bf16 logits may flip near-tie draws, so agreement is
    # statistical (the oracle, in fp32, reproduces this chain bit for bit: tests/test_oracle_golden.py)
    out = model.sample_one_batch(data=data, noise_fn=lambda c, shp: counter_noise(shp, c))
    gx, ge = out.X.argmax(-1).cpu().numpy(), out.E.argmax(-1).cpu().numpy()
    ax, ae = (gx == z['chain_X']).mean(), (ge == z['chain_E']).mean()
    print(f'config-0 chain vs reference: nodes {ax:.4f}, edges {ae:.4f}')
    # measured: nodes 1.0000, edges >= 0.998 (a handful of near-tie edge draws flip under bf16 logits and propagate)
    assert ax >= 0.98 and ae >= 0.99, (ax, ae)
    assert torch.equal(out.E, out.E.transpose(1, 2)) and bool((out.X.sum(-1) == 1).all()) and bool((out.E.sum(-1) == 1).all())


def test_chain_config0_fp32_inference_reproduces_the_reference_draw_for_draw():
    """The same chain through the fp32 inference mode (`model.model.inference_precision = 'fp32'`: the training forward's kernels
    under no_grad, every tensor and GEMM fp32, separate posterior / draw kernels): no bf16 near-tie flips, so the HIP path has to
    reproduce the reference's 10-step chain of configs[0] exactly, and its teacher-forced logits to fp32 rounding."""
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    z = load('chain_elbo_tiny')
    model = DiscreteDenoisingDiffusion(make_cfg(tiny_dims()), make_info(), device=D).eval()
    model.model.load_state_dict(weights_of(z))
    model.model.inference_precision = 'fp32'
    assert not model.model.supports_fused_step()
    am = T(z['am'])
    pos = S.shared_pos_encodings(dict(atom_map_numbers=am), 64, seed=73).to(D)
    model.pos_encodings_fn = lambda zt: pos
    data = PlaceHolder(X=F.one_hot(T(z['true_X']), KX).float().to(D), E=F.one_hot(T(z['true_E']), KE).float().to(D),
                       y=torch.zeros(4, 1, device=D), node_mask=T(z['node_mask']).to(D), atom_map_numbers=am.to(D),
                       mol_assignments=T(z['mol']).to(D))
    zt = data.get_new_object(X=F.one_hot(T(z['step_zt_X']), KX).float().to(D), E=F.one_hot(T(z['step_zt_E']), KE).float().to(D),
                             y=torch.full((4, 1), 6.0, device=D))
    with torch.no_grad():
        pred = model.forward(zt, pos_encodings=pos)
    rel = lambda a, b: ((a.cpu().double() - b.double()).norm() / b.double().norm()).item()
    rx, re_ = rel(pred.X, T(z['step_logit_X'])), rel(pred.E, T(z['step_logit_E']))
    print(f'fp32 inference, teacher-forced logits vs reference: rel-L2 nodes {rx:.2e}, edges {re_:.2e}')
    assert rx <= 1e-5 and re_ <= 1e-5, (rx, re_)
    out = model.sample_one_batch(data=data, noise_fn=lambda c, shp: counter_noise(shp, c))
    gx, ge = out.X.argmax(-1).cpu().numpy(), out.E.argmax(-1).cpu().numpy()
    assert np.array_equal(gx, z['chain_X']) and np.array_equal(ge, z['chain_E']), ((gx == z['chain_X']).mean(), (ge == z['chain_E']).mean())


def test_temperature_chain_and_recorded_chains_against_reference_golden():
    """tests/golden/chain_temp_tiny.npz: the reference's sample_one_batch(get_chains=True) at temperature_scaling_node = 0.7 /
    _edge = 1.4 with log_every_t = 3 (make_golden_r3.py).  The temperature enters the posterior of every step (da_posterior_sample's
    temp arguments) and the recorded pred_0 chain (softmax(T log_softmax(logits)), sampling.py:82,118-129).  Same structure of the
    recorded chains; at the first logged step (the state is still the reference's own: the limit draw is bit-exact) probabilities within
    the bf16 tolerance; final chain agreement statistical as for configs[0] (the oracle reproduces this chain bit for bit)."""
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    z, zt = load('chain_elbo_tiny'), load('chain_temp_tiny')
    cfg = make_cfg(tiny_dims())
    cfg.diffusion.temperature_scaling_node, cfg.diffusion.temperature_scaling_edge = (float(v) for v in zt['temps'])
    cfg.train.log_every_t = int(zt['log_every_t'])
    model = DiscreteDenoisingDiffusion(cfg, make_info(), device=D).eval()
    model.model.load_state_dict(weights_of(z))
    am = T(z['am'])
    pos = S.shared_pos_encodings(dict(atom_map_numbers=am), 64, seed=73).to(D)
    model.pos_encodings_fn = lambda zt_: pos
    data = PlaceHolder(X=F.one_hot(T(z['true_X']), KX).float().to(D), E=F.one_hot(T(z['true_E']), KE).float().to(D),
                       y=torch.zeros(4, 1, device=D), node_mask=T(z['node_mask']).to(D), atom_map_numbers=am.to(D),
                       mol_assignments=T(z['mol']).to(D))
    final, samples, probs, preds = model.sample_one_batch(data=data, get_chains=True, noise_fn=lambda c, shp: counter_noise(shp, c))
    assert [t for t, _ in samples] == [int(t) for t in zt['sample_steps']]
    assert [s for s, _ in probs] == [int(s) for s in zt['logged_steps']] == [s for s, _ in preds]
    # first logged step (s = 9): z_T is the reference's (bit-exact limit draw + apply_mask), so only the bf16 logits differ
    assert np.array_equal(samples[0][1].X.cpu().numpy(), zt['sample10_X']) and np.array_equal(samples[0][1].E.cpu().numpy(), zt['sample10_E'])
    for name, got, tol in (('prob9_X', probs[0][1].X, 3e-2), ('prob9_E', probs[0][1].E, 3e-2), ('pred9_X', preds[0][1].X, 3e-2),
                           ('pred9_E', preds[0][1].E, 3e-2)):
        err = float((got.cpu() - T(zt[name])).abs().max())
        assert err <= tol, (name, err)
    # the temperature must actually be in there: the same quantities at temperature 1 differ from the golden by far more
    cfg.diffusion.temperature_scaling_node = cfg.diffusion.temperature_scaling_edge = 1.0
    _, _, probs1, preds1 = model.sample_one_batch(data=data, get_chains=True, noise_fn=lambda c, shp: counter_noise(shp, c))
    assert float((preds1[0][1].E.cpu() - T(zt['pred9_E'])).abs().max()) > 0.1
    gx, ge = final.X.cpu().numpy(), final.E.cpu().numpy()
    ax, ae = (gx == zt['chain_X']).mean(), (ge == zt['chain_E']).mean()
    print(f'temperature chain vs reference: nodes {ax:.4f}, edges {ae:.4f}')
    assert ax >= 0.97 and ae >= 0.985, (ax, ae)
    # the fp32 inference mode: recorded probabilities to fp32 rounding and the chain draw for draw
    cfg.diffusion.temperature_scaling_node, cfg.diffusion.temperature_scaling_edge = (float(v) for v in zt['temps'])
    model.model.inference_precision = 'fp32'
    final, samples, probs, preds = model.sample_one_batch(data=data, get_chains=True, noise_fn=lambda c, shp: counter_noise(shp, c))
    for name, got in (('prob9_X', probs[0][1].X), ('prob9_E', probs[0][1].E), ('pred9_X', preds[0][1].X), ('pred9_E', preds[0][1].E)):
        err = float((got.cpu() - T(zt[name])).abs().max())
        assert err <= 2e-5, (name, err)
    assert np.array_equal(final.X.cpu().numpy(), zt['chain_X']) and np.array_equal(final.E.cpu().numpy(), zt['chain_E'])


def test_sampling_properties_at_max_atoms():
    """n = 300 (MAX_ATOMS_RXN), full dims, 2 layers, 3 steps: condition untouched, symmetric one-hot output, and the
    same seed gives the same chain (determinism of the whole HIP path)."""
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    dims = dict(n_layers=2, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
                hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
    torch.manual_seed(0)
    model = DiscreteDenoisingDiffusion(make_cfg(dims, T_=500, T_eval=5, extra=True), make_info(), device=D).eval()
    B, n = 2, 300
    true = S.make_reaction_batch(B, n, seed=5, P=None)
    pos = S.shared_pos_encodings(true, 256, seed=6).to(D)
    model.pos_encodings_fn = lambda zt: pos
    g = torch.Generator().manual_seed(7)
    xe, ye = torch.rand(B, n, 8, generator=g).to(D), torch.rand(B, 12, generator=g).to(D)
    model.extra_features_fn = lambda zt: (xe, ye)
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    outs = []
    for _ in range(2):
        outs.append(model.sample_one_batch(data=data, noise_fn=lambda c, shp: S.exp_noise(shp, 100 + c)))
    a, b = outs
    assert torch.equal(a.X, b.X) and torch.equal(a.E, b.E)
    assert torch.equal(a.E, a.E.transpose(1, 2))
    assert bool((a.X.sum(-1) == 1).all()) and bool((a.E.sum(-1) == 1).all())
    cls = true['X'].argmax(-1)
    ar = torch.arange(n)[None].expand(B, n)
    spos = torch.where(cls == 28, ar, torch.full_like(ar, -1)).max(-1).values
    fixed = ar >= spos[:, None]                      # SuNo, product, padding: never touched
    assert torch.equal(a.X.cpu()[fixed], true['X'][fixed])
    fe = fixed[:, :, None] | fixed[:, None, :]
    assert torch.equal(a.E.cpu()[fe], true['E'][fe])
    pad = ~true['node_mask']
    assert bool((a.X.cpu()[pad].argmax(-1) == 0).all())


def test_marginal_sampling_at_twice_max_atoms():
    """BASELINE.json configs[4]: transition='marginal' (dense marginal Qbar_t posterior), n = 600 (the edge-tensor
    bandwidth stress), full dims, 2 layers, 3 steps: same size-independent properties as above."""
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    dims = dict(n_layers=2, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
                hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
    torch.manual_seed(0)
    cfg = make_cfg(dims, T_=500, T_eval=5, extra=True)
    cfg.diffusion.transition = 'marginal'
    info = make_info()
    g = torch.Generator().manual_seed(11)
    info.node_types = torch.rand(KX, generator=g) + 0.1       # a fixed seeded probability vector over the classes
    info.edge_types = torch.rand(KE, generator=g) + 0.1
    model = DiscreteDenoisingDiffusion(cfg, info, device=D).eval()
    B, n = 1, 600
    true = S.make_reaction_batch(B, n, seed=9, P=None)
    pos = S.shared_pos_encodings(true, 256, seed=6).to(D)
    model.pos_encodings_fn = lambda zt: pos
    xe, ye = torch.rand(B, n, 8, generator=g).to(D), torch.rand(B, 12, generator=g).to(D)
    model.extra_features_fn = lambda zt: (xe, ye)
    data = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
    outs = [model.sample_one_batch(data=data, noise_fn=lambda c, shp: S.exp_noise(shp, 300 + c)) for _ in range(2)]
    a, b = outs
    assert torch.equal(a.X, b.X) and torch.equal(a.E, b.E)
    assert torch.equal(a.E, a.E.transpose(1, 2))
    assert bool((a.X.sum(-1) == 1).all()) and bool((a.E.sum(-1) == 1).all())
    cls = true['X'].argmax(-1)
    ar = torch.arange(n)[None].expand(B, n)
    spos = torch.where(cls == 28, ar, torch.full_like(ar, -1)).max(-1).values
    fixed = ar >= spos[:, None]
    assert torch.equal(a.X.cpu()[fixed], true['X'][fixed])
    fe = fixed[:, :, None] | fixed[:, None, :]
    assert torch.equal(a.E.cpu()[fe], true['E'][fe])
    # masked states of the marginal family are never drawn on free nodes / pairs (noise_schedule.py:88-134)
    free = ~fixed & true['node_mask']
    assert not bool(((a.X.cpu().argmax(-1) == 28) & free).any()) and not bool(((a.X.cpu().argmax(-1) == 0) & free).any())
    fee = free[:, :, None] & free[:, None, :]
    assert not bool((a.E.cpu().argmax(-1)[fee] >= 4).any())


def test_config2_shape_100_steps_100_samples_one_condition():
    """BASELINE.json configs[2] on one GPU: sample_for_condition with 100 samples of one condition at n = 300, the full 9-layer
    align_absorbing denoiser, T = 500 / 100 evaluation steps, extra features and Laplacian positional encodings computed on the
    device (nothing is an injected input).  No reference run exists at this size (no checkpoint offline), so the checks are the
    size-independent properties: collapsed symmetric output, the condition untouched, padding stays class 0, chains with
    different seeds differ, and the same seeds give the same 100 chains in a second run."""
    from diffalign_amd.diffusion_rxn import DiscreteDenoisingDiffusionRxn
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd.distributed import chain_seed
    from diffalign_amd import synthetic as S
    from tests.golden.common import extra_info
    dims = dict(n_layers=9, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
                hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
    torch.manual_seed(0)
    cfg = make_cfg(dims, T_=500, T_eval=100, extra=True)
    cfg.neuralnet.num_lap_eig_vectors = 20
    cfg.train.log_every_t = 50
    info = make_info()
    for k, v in vars(extra_info(ns)).items():
        setattr(info, k, v)
    model = DiscreteDenoisingDiffusionRxn(cfg, info, device=D).eval()
    n, n_samples = 300, 100
    true = S.make_reaction_batch(1, n, seed=17, P=30)                     # a USPTO-sized product (30 atoms) padded to MAX_ATOMS_RXN
    seeds = torch.tensor([chain_seed(1, g) for g in range(n_samples)], dtype=torch.int64)
    runs = []
    for _ in range(2):
        torch.manual_seed(123)          # the random signs of the Laplacian encodings and the eigen-features' noise on padded slots
        cond = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
        runs.append(model.sample_for_condition(dense_data=cond, n_samples=n_samples, inpaint_node_idx=None, inpaint_edge_idx=None,
                                               chain_seeds=seeds))
    a, b = runs
    assert a.X.shape == (n_samples, n) and a.E.shape == (n_samples, n, n)
    assert torch.equal(a.X, b.X) and torch.equal(a.E, b.E)
    assert torch.equal(a.E, a.E.transpose(1, 2))
    cls, ecl = true['X'][0].argmax(-1), true['E'][0].argmax(-1)
    s = int((cls == 28).nonzero()[0])
    assert torch.equal(a.X.cpu()[:, s:], cls[s:].expand(n_samples, -1))
    assert torch.equal(a.E.cpu()[:, s:, :], ecl[s:, :].expand(n_samples, -1, -1))
    assert bool((a.X.cpu()[:, ~true['node_mask'][0]] == 0).all())
    assert len({tuple(r.tolist()) for r in a.X.cpu()[:, :s]}) > 1         # chains are not copies of each other


def test_graphed_chain_equals_eager_chain():
    """graphs.GraphedChain: every reverse step of a small batch replayed as a HIP graph (three-stream arrangement captured, extra
    features with the eigensolver's fixed-block cache inside, counter noise, inpainting) gives the eager loop's samples bit for bit;
    a second chain of the same shape reuses the graphs (nothing is captured again) and a chain of another shape gets its own."""
    from diffalign_amd.diffusion_rxn import DiscreteDenoisingDiffusionRxn
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd.distributed import chain_seed
    from diffalign_amd import synthetic as S, graphs
    from tests.golden.common import extra_info
    dims = dict(n_layers=2, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
                hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
    torch.manual_seed(0)
    cfg = make_cfg(dims, T_=20, T_eval=10, extra=True)
    cfg.neuralnet.pos_encoding_type = 'smiles_pos_enc'
    cfg.train.log_every_t = 3
    info = make_info()
    for k, v in vars(extra_info(ns)).items():
        setattr(info, k, v)
    model = DiscreteDenoisingDiffusionRxn(cfg, info, device=D).eval()
    old = graphs.CHAIN_GRAPHS
    try:
        for case, (n, P, B, inp) in enumerate([(40, None, 5, True), (40, None, 5, False), (33, 8, 3, False)]):
            true = S.make_reaction_batch(B, n, seed=30 + case, P=P)
            seeds = torch.tensor([chain_seed(2, 10 * case + g) for g in range(B)], dtype=torch.int64)
            kw = dict(inpaint_node_idx=[[0, 1]] + [[] for _ in range(B - 1)] if inp else None,
                      inpaint_edge_idx=[[[0, 1]]] + [[] for _ in range(B - 1)] if inp else None, chain_seeds=seeds)
            res = {}
            for mode in ('0', '1'):
                graphs.CHAIN_GRAPHS = mode
                torch.manual_seed(77)                                       # the per-chain noise on padded slots
                cond = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
                res[mode] = model.sample_for_condition(dense_data=cond, n_samples=1, **kw)
            assert torch.equal(res['0'].X, res['1'].X) and torch.equal(res['0'].E, res['1'].E), case
            assert not torch.equal(res['1'].X[0], res['1'].X[1])
        cache = model._chain_graphs
        assert len(cache) == 3 and all(len(g.graphs) == 10 for g in cache.values())
        graphs.CHAIN_GRAPHS = '1'
        first = next(iter(cache.values()))
        ids = [id(g) for g in first.graphs.values()]
        true = S.make_reaction_batch(5, 40, seed=99, P=None)
        cond = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
        seeds = torch.arange(5, dtype=torch.int64) + 1000
        torch.manual_seed(5)
        a = model.sample_for_condition(dense_data=cond, n_samples=1, inpaint_node_idx=[[2, 3]] + [[]] * 4, inpaint_edge_idx=[[[1, 2]]] + [[]] * 4, chain_seeds=seeds)
        assert [id(g) for g in first.graphs.values()] == ids              # replayed, not re-captured
        graphs.CHAIN_GRAPHS = '0'
        torch.manual_seed(5)
        cond = PlaceHolder(**{k: v.to(D) for k, v in true.items()})
        b = model.sample_for_condition(dense_data=cond, n_samples=1, inpaint_node_idx=[[2, 3]] + [[]] * 4, inpaint_edge_idx=[[[1, 2]]] + [[]] * 4, chain_seeds=seeds)
        assert torch.equal(a.X, b.X) and torch.equal(a.E, b.E)
    finally:
        graphs.CHAIN_GRAPHS = old
