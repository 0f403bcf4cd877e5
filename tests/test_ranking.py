"""CPU: ELBO ranking driver host logic (diffalign_amd/diffusion_rxn.py) against the reference's outputs
(tests/golden/ranking.npz: sort_by_elbo, diffusion_rxn.py:643-699, and reactions_sorted_with_weighted_prob, graph.py:709-736)."""
import json

import torch

from tests.golden.common import load


def test_sort_by_elbo_and_weighted_ranking_match_reference():
    from diffalign_amd.diffusion_rxn import DiscreteDenoisingDiffusionRxn as Rxn, reactions_sorted_with_weighted_prob
    z = load('ranking')
    T = lambda k: torch.from_numpy(z[k])
    is_unique = [bool(v) for v in z['is_unique']]
    counts = [int(v) for v in z['counts']]
    bs, ns_ = int(z['bs']), int(z['n_samples'])
    u = [i for i, f in enumerate(is_unique) if f]
    rct = [[f'R{i}a', f'R{i}b'] for i in u]
    prod = [[f'P{i // ns_}'] for i in u]
    res = Rxn.sort_by_elbo(None, T('elbos'), T('loss_t'), T('loss_0'), rct, prod, is_unique, counts, bs, ns_)
    assert json.loads(json.dumps(res, sort_keys=True)) == json.loads(bytes(z['sorted_json']).decode())
    ranked = reactions_sorted_with_weighted_prob(res, 0.3)
    want = json.loads(bytes(z['ranked_json']).decode())
    got = json.loads(json.dumps(ranked, sort_keys=True))
    assert got.keys() == want.keys()
    for k in want:
        assert [r['sample_idx'] for r in got[k]] == [r['sample_idx'] for r in want[k]]
        for a, b in zip(got[k], want[k]):
            assert abs(a['weighted_prob'] - b['weighted_prob']) <= 1e-12 and a['rcts'] == b['rcts'] and a['count'] == b['count']
