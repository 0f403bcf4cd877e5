"""GPU: the user API of diffalign/inference.py and diffalign/model.py on a synthetic dense condition.  Chemistry
(SMILES <-> graph) is an injected callable: here the 'SMILES' is a key into a table of synthetic reactions."""
import numpy as np
import pytest
import torch

from tests.golden.common import load, weights_of, tiny_dims, ATOMS, BONDS
from tests.test_gpu_sampling import make_cfg, make_info

pytestmark = pytest.mark.gpu
D = 'cuda:0'


@pytest.fixture()
def api():
    import diffalign_amd
    from diffalign_amd import inference, synthetic as S
    from diffalign_amd.diffusion_rxn import DiscreteDenoisingDiffusionRxn
    from diffalign_amd.placeholder import PlaceHolder
    z = load('chain_elbo_tiny')
    cfg = make_cfg(tiny_dims(), T_=10, T_eval=1)
    cfg.neuralnet.pos_encoding_type = 'smiles_pos_enc'
    model = DiscreteDenoisingDiffusionRxn(cfg, make_info(), device=D).eval()
    model.model.load_state_dict(weights_of(z))
    table = {f'P{k}': {key: v[k:k + 1] for key, v in S.make_reaction_batch(3, 24, seed=72).items()} for k in range(3)}
    calls = []

    def encoder(smiles, cfg_):
        calls.append(smiles)
        return PlaceHolder(**{k: v.clone() for k, v in table[smiles].items()})

    inference.configure(model=model, cfg=cfg, encoder=encoder, validator=lambda s: not str(s).startswith('bad'))
    yield diffalign_amd, inference, model, cfg, table, calls
    inference._cached_model = inference._cached_cfg = None
    inference._hooks.update(encoder=None, decoder=None, validator=None)


def test_predict_precursors_and_transition_cache(api):
    pkg, inference, model, cfg, table, calls = api
    assert pkg.predict_precursors('bad smiles') == [] and calls == []
    with pytest.raises(AssertionError, match='must divide'):
        pkg.predict_precursors('P0', n_precursors=2, diffusion_steps=3)            # 3 does not divide T = 10
    res = pkg.predict_precursors('P0', n_precursors=6, diffusion_steps=5)
    assert cfg.diffusion.diffusion_steps_eval == 5 and model.transition_model_eval.Qts_x.shape[0] == 6
    tm5 = model.transition_model_eval
    assert 1 <= len(res) <= 6 and abs(sum(r['score'] for r in res) - 1.0) < 1e-6 or len(res) == 6
    assert all(set(r) == {'precursors', 'score', 'sample_data', 'atom_mapping'} for r in res)
    assert res == sorted(res, key=lambda r: r['score'], reverse=True)
    sd = res[0]['sample_data']
    X = torch.tensor(sd['X'])
    cls = table['P0']['X'][0].argmax(-1)
    s = int((cls == 28).nonzero()[0])
    assert X.shape == (1, 24) and torch.equal(X[0, s:], cls[s:])                  # SuNo + product + padding: the condition's
    pkg.predict_precursors('P1', n_precursors=2, diffusion_steps=2)
    pkg.predict_precursors('P1', n_precursors=2, diffusion_steps=5)
    assert model.transition_model_eval is tm5                                     # cached per step count


def test_predict_with_inpainting_keeps_nodes_and_reports_failure(api):
    pkg, inference, model, cfg, table, calls = api
    first = pkg.predict_precursors('P2', n_precursors=4, diffusion_steps=5)[0]
    prev = first['sample_data']
    X0 = torch.tensor(prev['X'])[0]
    real = [i for i in range(24) if ATOMS[int(X0[i])] not in ('none', 'U', 'Au', 'SuNo') and i < 5]
    keep = real[:2]
    results, failure = pkg.predict_with_inpainting('P2', prev, keep, n_precursors=8, diffusion_steps=5)
    assert (failure is None) != (results == [])
    for r in results:
        Xn = torch.tensor(r['sample_data']['X'])[0]
        assert torch.equal(Xn[keep], X0[keep])                                       # the kept atoms are the previous sample's
    if failure is not None:
        assert failure['n_samples'] == 8 and all(a['index'] not in keep for a in failure['requested_change_atoms'])
    # keeping every real atom leaves nothing to change: the constraint check is skipped and all samples come back
    everything = [i for i in range(24) if bool(torch.tensor(prev['node_mask'])[0][i])]
    results, failure = pkg.predict_with_inpainting('P2', prev, everything, n_precursors=3, diffusion_steps=5)
    assert failure is None and len(results) >= 1


def test_diffalign_model_adapter(api):
    pkg, inference, model, cfg, table, calls = api
    m = pkg.DiffAlignModel(diffusion_steps=2, samples_per_product=5, device=D)
    assert m.name == 'DiffAlign' and m.get_default_model_dir().name == 'checkpoints'
    assert sum(p.numel() for p in m.get_parameters()) == sum(p.numel() for p in model.parameters())
    out = m._get_reactions(['P0', 'P1'], num_results=3)
    assert len(out) == 2 and all(1 <= len(o) <= 3 for o in out)
    for reactions in out:
        probs = [meta['probability'] for _, meta in reactions]
        assert probs == sorted(probs, reverse=True) and all(0 < p <= 1 for p in probs)
    with pytest.raises(AssertionError, match='must divide'):
        pkg.DiffAlignModel(diffusion_steps=4, samples_per_product=2, device=D)._get_reactions(['P0'], 1)
