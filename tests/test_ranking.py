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


def test_scoring_list_logic_matches_reference():
    """tests/golden/scoring_lists.npz (make_golden_r3.py): the reference's get_unique_indices_from_reaction_list,
    split_reactions_to_reactants_and_products and calculate_top_k (utils/graph.py:160-187, 738-770) on 3 conditions x 5 synthetic
    reaction strings with duplicates, permuted reactants and a dummy '>>product' sample, through sort_by_elbo and the weighted ranking."""
    from types import SimpleNamespace as ns
    from diffalign_amd import graph
    from diffalign_amd.diffusion_rxn import DiscreteDenoisingDiffusionRxn as Rxn, reactions_sorted_with_weighted_prob
    from tests.golden.make_golden_r3_strings import scoring_strings
    z = load('scoring_lists')
    gen, true = scoring_strings()
    ui, counts, is_unique = graph.get_unique_indices_from_reaction_list(gen)
    assert ui == [int(v) for v in z['unique_indices']] and is_unique == [bool(v) for v in z['is_unique']]
    gen_u = [r for r, u in zip(gen, is_unique) if u]
    rcts, prods = graph.split_reactions_to_reactants_and_products(gen_u)
    assert [rcts, prods] == json.loads(bytes(z['split_json']).decode())
    counts = list(counts)
    elbos = torch.from_numpy(z['elbos']).clone()
    for i, rct in enumerate(rcts):
        if all(a == '' for a in rct):
            assert torch.isinf(elbos[i]); counts[i] = 0
    assert counts == [int(v) for v in z['counts']]
    cfg = ns(test=ns(topks=[1, 2, 3], sort_lambda_value=0.3))
    res = Rxn.sort_by_elbo(None, elbos, torch.from_numpy(z['loss_t']), torch.from_numpy(z['loss_0']), rcts, prods, is_unique, counts, 3, 5)
    assert json.loads(json.dumps(res, sort_keys=True)) == json.loads(bytes(z['sorted_json']).decode())
    ranked = reactions_sorted_with_weighted_prob(res, 0.3)
    t_r, t_p = graph.split_reactions_to_reactants_and_products(true)
    topk, mrr = graph.calculate_top_k(cfg, res, t_r[::5], t_p[::5])
    topk_w, mrr_w = graph.calculate_top_k(cfg, ranked, t_r[::5], t_p[::5])
    assert torch.equal(topk, torch.from_numpy(z['topk'])) and torch.equal(mrr, torch.from_numpy(z['mrr']))
    assert torch.equal(topk_w, torch.from_numpy(z['topk_weighted'])) and torch.equal(mrr_w, torch.from_numpy(z['mrr_weighted']))


def test_sparse_fields_and_text_formats(tmp_path):
    """SURVEY 8 f-4 (what can be built without torch_geometric / rdkit): the per-graph fields of PlaceHolder.pyg()
    (utils/placeholder.py:165-179, dense_to_sparse order and dtypes) and the samples_*.txt layout (utils/graph.py:786-800)."""
    import numpy as np
    import torch
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import graph
    g = torch.Generator().manual_seed(0)
    B, n = 3, 7
    E = torch.randint(0, 4, (B, n, n), generator=g); E = torch.triu(E, 1); E = E + E.transpose(1, 2)
    X = torch.randint(0, 9, (B, n), generator=g)
    ph = PlaceHolder(X=X, E=E, y=torch.zeros(B, 1), node_mask=torch.ones(B, n, dtype=torch.bool), atom_map_numbers=torch.arange(n).repeat(B, 1),
                     mol_assignments=torch.ones(B, n, dtype=torch.long))
    fields = ph.to_sparse()
    one_hot = PlaceHolder(X=torch.nn.functional.one_hot(X, 9).float(), E=torch.nn.functional.one_hot(E, 4).float(), y=ph.y, node_mask=ph.node_mask,
                          atom_map_numbers=ph.atom_map_numbers, mol_assignments=ph.mol_assignments).to_sparse()
    for i, d in enumerate(fields):
        rows, cols = np.nonzero(E[i].numpy())                       # row-major order, as dense_to_sparse
        assert d['edge_index'].dtype == torch.int16 and d['edge_attr'].dtype == torch.uint8 and d['x'].dtype == torch.uint8
        assert np.array_equal(d['edge_index'].numpy(), np.stack([rows, cols])) and np.array_equal(d['edge_attr'].numpy(), E[i].numpy()[rows, cols])
        assert np.array_equal(d['x'].numpy(), X[i].numpy()) and d['mask_atom_mapping'].tolist() == list(range(n))
        assert all(torch.equal(d[k], one_hot[i][k]) for k in ('x', 'edge_index', 'edge_attr'))
    f = str(tmp_path / 'samples.txt')
    graph.save_gen_rxn_smiles_to_file(f, 0, ['A.B>>C', 'D>>C'], 'A.B>>C')
    graph.save_gen_rxn_smiles_to_file(f, 1, ['E>>F'], 'G>>F')
    assert open(f).read() == '(cond 0) A.B>>C:\n\tA.B>>C\n\tD>>C\n(cond 1) G>>F:\n\tE>>F\n'
    f2 = str(tmp_path / 'eval.txt')
    graph.save_samples_to_file_without_weighted_prob(f2, 4, {'a': [dict(rxn='A>>B', elbo=1.5, loss_t=1.0, loss_0=0.5, count=3)]}, ['A>>B'])
    assert open(f2).read() == "(cond 4) A>>B:\n\t('A>>B', [1.5, 1.0, 0.5, 3])\n"


def test_statistics_tail_policy_and_fixed_block_of_the_eigen_cache():
    """Host logic of two round-2 switches.  engine.tail_chunks / use_stats_tail: about 1,300 pairs per ticket, between 4 and 64 tickets
    per graph, the tail from n = 112 on.  EigCache.from_free_masks: the fixed block of a graph starts behind its last node that is real
    and free to diffuse (padding and condition nodes behind it belong to the block); sizes and state of the cache."""
    import torch
    from diffalign_amd import engine
    from diffalign_amd.extra_features import EigCache
    if engine._TAIL_CHUNKS_ENV is None and engine._STATS_TAIL_ENV is None:
        assert [engine.tail_chunks(n) for n in (20, 72, 140, 300, 1000)] == [4, 4, 15, 64, 64]
        assert [engine.use_stats_tail(n) for n in (72, 111, 112, 300)] == [False, False, True, True]
    node_mask = torch.tensor([[1, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1, 1], [1, 1, 0, 0, 0, 0, 0]], dtype=torch.bool)
    free_e = torch.tensor([[1, 1, 0, 0, 0, 0, 0], [1, 0, 1, 0, 0, 0, 1], [0, 0, 1, 1, 0, 0, 0]], dtype=torch.bool)
    c = EigCache.from_free_masks(free_e, node_mask)
    assert c.fixed_from.tolist() == [2, 7, 0]          # graph 1: a free node at the very end, nothing to keep; graph 2: nothing free
    assert c.mmax == 7 and c.store.shape == (3, 9, 7) and c.state.tolist() == [0, 0, 0]
