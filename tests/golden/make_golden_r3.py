"""Round-3 golden vectors from the REFERENCE (same set-up as make_golden.py: read-only checkout at /root/reference imported in the build
container, third-party modules it lacks stubbed, torch.multinomial replaced by the exponential race on explicit counter noise).

  chain_temp_tiny.npz   BASELINE.json configs[0]'s tiny model, 10-step chain with temperature_scaling_node = 0.7 and
                        temperature_scaling_edge = 1.4 (diffusion/sampling.py:82), sample_one_batch(get_chains=True) with
                        log_every_t = 3: the final sample, and what the reference records for plotting at the logged steps -- the state
                        z_t, p(z_s | z_t) after apply_mask (:131-144) and the denoiser's prediction as the reference turns it into
                        probabilities, softmax(temperature * log_softmax(logits)) with the supernode class dropped (:118-129).
                        Pins the temperature path, which every other fixture runs at 1.0.

  scoring_lists.npz     the list logic of the scoring driver on synthetic canonical reaction strings (3 conditions x 5 samples with
                        duplicates, permuted reactants and a dummy '>>product' sample): get_unique_indices_from_reaction_list,
                        split_reactions_to_reactants_and_products (utils/graph.py:160-187, 767-770), then sort_by_elbo +
                        reactions_sorted_with_weighted_prob on seeded scores and calculate_top_k (:738-765) of both rankings.

Run:  CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" python tests/golden/make_golden_r3.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import make_golden as G     # noqa: E402  (registers the stubs, imports the reference)

import torch                                   # noqa: E402

TEMP_NODE, TEMP_EDGE, LOG_EVERY = 0.7, 1.4, 3


def gen_temperature_chain():
    dims = G.tiny_dims()
    cfg = G.make_cfg(dims, T=10, T_eval=10)
    cfg.diffusion.temperature_scaling_node, cfg.diffusion.temperature_scaling_edge = TEMP_NODE, TEMP_EDGE
    cfg.train.log_every_t = LOG_EVERY
    model, sd = G.build_diffusion(cfg, seed=71)          # the weights of chain_elbo_tiny.npz
    true = G.S.make_reaction_batch(4, 24, seed=72)
    model._orig_am = true['atom_map_numbers'].clone()
    out = {}
    torch.manual_seed(0)
    torch.Tensor.multinomial = G._race_multinomial
    try:
        G._noise_state['counter'] = 0
        with torch.no_grad():
            final, sample_chains, prob_s_chains, pred_0_chains = model.sample_one_batch(device='cpu', data=G.ph(true), get_chains=True)
        out['noise_calls'] = G._noise_state['counter']
    finally:
        torch.Tensor.multinomial = G._real_multinomial
    out['chain_X'], out['chain_E'] = final.X, final.E                       # collapsed class indices (mask(collapse=True))
    out['sample_steps'] = torch.tensor([t for t, _ in sample_chains])
    out['logged_steps'] = torch.tensor([s for s, _ in prob_s_chains])
    for (t, smp) in sample_chains:
        out[f'sample{t}_X'], out[f'sample{t}_E'] = smp.X, smp.E
    for (s, pr), (s2, p0) in zip(prob_s_chains, pred_0_chains):
        assert s == s2
        out[f'prob{s}_X'], out[f'prob{s}_E'] = pr.X, pr.E
        out[f'pred{s}_X'], out[f'pred{s}_E'] = p0.X, p0.E
    out['temps'] = torch.tensor([TEMP_NODE, TEMP_EDGE])
    out['log_every_t'] = torch.tensor(LOG_EVERY)
    G.save('chain_temp_tiny', **out)
    print('logged steps', out['logged_steps'].tolist(), 'sample steps', out['sample_steps'].tolist(), 'noise calls', out['noise_calls'])


from tests.golden.make_golden_r3_strings import scoring_strings   # noqa: E402


def gen_scoring_lists():
    import json
    import numpy as np
    from diffalign.diffusion.diffusion_rxn import DiscreteDenoisingDiffusionRxn as RefRxn
    from diffalign.utils import graph as ref_graph
    gen, true = scoring_strings()
    bs, ns_ = 3, 5
    unique_indices, counts, is_unique = ref_graph.get_unique_indices_from_reaction_list(gen)
    gen_u = [r for r, u in zip(gen, is_unique) if u]
    rcts, prods = ref_graph.split_reactions_to_reactants_and_products(gen_u)
    g = torch.Generator().manual_seed(21)
    elbos = torch.rand(len(gen_u), generator=g) * 40 + 5
    loss_t = torch.rand(len(gen_u), generator=g) * 30
    loss_0 = -torch.rand(len(gen_u), generator=g)
    counts = [int(c) for c in counts]
    for i, rct in enumerate(rcts):                       # diffusion_rxn.py:459-463
        if len([a for a in rct if a == '']) == len(rct):
            elbos[i] = float('inf'); counts[i] = 0
    cfg = G.ns(test=G.ns(topks=[1, 2, 3], sort_lambda_value=0.3))
    res = RefRxn.sort_by_elbo(None, elbos, loss_t, loss_0, rcts, prods, is_unique, counts, bs, ns_, k=cfg.test.topks)
    ranked = ref_graph.reactions_sorted_with_weighted_prob(res, 0.3)
    t_r, t_p = ref_graph.split_reactions_to_reactants_and_products(true)
    t_r, t_p = t_r[::ns_], t_p[::ns_]
    topk, mrr = ref_graph.calculate_top_k(cfg, res, t_r, t_p)
    topk_w, mrr_w = ref_graph.calculate_top_k(cfg, ranked, t_r, t_p)
    js = lambda o: np.frombuffer(json.dumps(o, sort_keys=True).encode(), dtype=np.uint8)
    G.save('scoring_lists', unique_indices=torch.tensor([int(i) for i in unique_indices]), counts=torch.tensor(counts),
           is_unique=torch.tensor([bool(u) for u in is_unique]), elbos=elbos, loss_t=loss_t, loss_0=loss_0, topk=topk, mrr=mrr,
           topk_weighted=topk_w, mrr_weighted=mrr_w, split_json=js([rcts, prods]), sorted_json=js(res), ranked_json=js(ranked))
    print('unique', [int(i) for i in unique_indices], 'counts', counts, 'topk', topk.tolist(), 'mrr', mrr.tolist())


if __name__ == '__main__':
    torch.set_num_threads(8)
    G.check_race_equals_multinomial()
    which = sys.argv[1:] or ['temperature', 'scoring']
    if 'temperature' in which: gen_temperature_chain()
    if 'scoring' in which: gen_scoring_lists()
    print('done')
