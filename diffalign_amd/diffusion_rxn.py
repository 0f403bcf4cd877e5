"""ELBO ranking driver: the part of `DiscreteDenoisingDiffusionRxn` (diffusion/diffusion_rxn.py) that scores
generated samples with the variational bound and ranks them.  The bound itself runs through the HIP kernels
(`ELBOMixin.elbo`: denoiser, k_elbo_rows, k_masked_sums); ranking is host logic over per-sample scalars and SMILES strings,
as in the reference.  SMILES conversion (rdkit) stays outside this package: `sort_by_elbo` takes the strings as inputs."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .diffusion_abstract import DiscreteDenoisingDiffusion


def reactions_sorted_with_weighted_prob(restructured_data, lambda_value):
    """utils/graph.py:709-736: per product, weighted_prob = lambda * softmax(-elbo) + (1 - lambda) * count share; sort descending."""
    restructured_data = copy.deepcopy(restructured_data)
    for product, reactions_list in restructured_data.items():
        sum_exp_elbo = sum(np.exp(-reaction['elbo']) for reaction in reactions_list)
        sum_counts = sum(reaction['count'] for reaction in reactions_list)
        for reaction in reactions_list:
            exp_elbo = np.exp(-reaction['elbo'])
            reaction['weighted_prob'] = (exp_elbo / sum_exp_elbo) * lambda_value + (reaction['count'] / sum_counts) * (1 - lambda_value)
        restructured_data[product] = sorted(reactions_list, key=lambda x: x['weighted_prob'], reverse=True)
    return restructured_data


class DiscreteDenoisingDiffusionRxn(DiscreteDenoisingDiffusion):
    """diffusion/diffusion_rxn.py: scoring and ranking of samples (the chemistry-side methods -- SMILES, plots, dataset
    evaluation -- are out of scope)."""

    @torch.no_grad()
    def estimate_elbo_with_repeats(self, final_samples_one_hot, noise_fn=None):
        """diffusion_rxn.py:606-641: mean over cfg.test.repeat_elbo estimates of the per-sample bound, the batch cut into chunks
        of cfg.test.elbo_batch_size samples.  Everything stays on the device (the reference reduces on the CPU)."""
        chunk_size = getattr(self.cfg.test, 'elbo_batch_size', None)
        total = final_samples_one_hot.X.shape[0]
        if chunk_size is None or chunk_size <= 0 or chunk_size >= total:
            chunk_ranges = [(0, total)]
        else:
            chunk_ranges = [(s, min(s + chunk_size, total)) for s in range(0, total, chunk_size)]
        reps = [[], [], []]
        for _ in range(self.cfg.test.repeat_elbo):
            chunks = [[], [], []]
            for start, end in chunk_ranges:
                inp = final_samples_one_hot if len(chunk_ranges) == 1 else final_samples_one_hot.subset_by_idx(start, end)
                out = self.elbo(dense_true=inp, avg_over_batch=False, noise_fn=noise_fn)
                for dst, v in zip(chunks, out):
                    dst.append(v.float())
            for dst, c in zip(reps, chunks):
                dst.append(torch.cat(c, dim=0).unsqueeze(0))
        return tuple(torch.cat(r, dim=0).mean(0) for r in reps)          # elbos, loss_t, loss_0: (bs * n_samples,)

    @torch.no_grad()
    def sort_by_elbo(self, elbos, loss_t, loss_0, rct_smiles, prod_smiles, is_unique, counts, bs, n_samples, k=[10]):
        """diffusion_rxn.py:643-699: per condition, the unique samples sorted by ascending bound, as
        {product: [{'rcts','prod','elbo','loss_t','loss_0','sample_idx','count'}, ...]}."""
        elbos, loss_0 = elbos.detach().cpu(), loss_0.detach().cpu()
        loss_t = loss_t.detach().cpu() if torch.is_tensor(loss_t) else torch.zeros_like(elbos)
        unique_indices = torch.tensor(is_unique).nonzero(as_tuple=True)[0]
        assert len(rct_smiles) == sum(is_unique) and len(prod_smiles) == sum(is_unique), 'sample_smiles is different than expected.'
        counts_t = torch.tensor(counts)
        rcts_all = np.empty(len(rct_smiles), dtype=object)
        prods_all = np.empty(len(prod_smiles), dtype=object)
        for i in range(len(rct_smiles)):
            rcts_all[i] = rct_smiles[i]; prods_all[i] = prod_smiles[i]
        gen_rxns = {}
        for i in range(bs):
            sel = np.array(((unique_indices >= i * n_samples) & (unique_indices < (i + 1) * n_samples)).nonzero(as_tuple=True)[0])
            e_sorted, idx_sorted = elbos[sel].sort(-1)
            lt, l0, cn = loss_t[sel][idx_sorted], loss_0[sel][idx_sorted], counts_t[sel][idx_sorted]
            rcts_, prods_ = rcts_all[sel], prods_all[sel]
            rows = [{'rcts': rcts_[s], 'prod': prods_[s], 'elbo': e_sorted[j].item(), 'loss_t': lt[j].item(), 'loss_0': l0[j].item(),
                     'sample_idx': s.item(), 'count': cn[j].item()} for j, s in enumerate(idx_sorted)]
            product = '.'.join(prods_[0])
            gen_rxns.setdefault(product, []).extend(rows)
        return gen_rxns
