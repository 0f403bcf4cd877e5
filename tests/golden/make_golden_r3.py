"""Round-3 golden vectors from the REFERENCE (same set-up as make_golden.py: read-only checkout at /root/reference imported in the build
container, third-party modules it lacks stubbed, torch.multinomial replaced by the exponential race on explicit counter noise).

  chain_temp_tiny.npz   BASELINE.json configs[0]'s tiny model, 10-step chain with temperature_scaling_node = 0.7 and
                        temperature_scaling_edge = 1.4 (diffusion/sampling.py:82), sample_one_batch(get_chains=True) with
                        log_every_t = 3: the final sample, and what the reference records for plotting at the logged steps -- the state
                        z_t, p(z_s | z_t) after apply_mask (:131-144) and the denoiser's prediction as the reference turns it into
                        probabilities, softmax(temperature * log_softmax(logits)) with the supernode class dropped (:118-129).
                        Pins the temperature path, which every other fixture runs at 1.0.

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


if __name__ == '__main__':
    torch.set_num_threads(8)
    G.check_race_equals_multinomial()
    gen_temperature_chain()
    print('done')
