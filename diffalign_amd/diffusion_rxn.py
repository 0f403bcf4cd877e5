"""Sampling drivers (`sample_for_condition`, `sample_n_conditions`: diffusion/diffusion_rxn.py:31-128, BASELINE configs[2]) and the
ELBO ranking driver: the part of `DiscreteDenoisingDiffusionRxn` (diffusion/diffusion_rxn.py) that scores
generated samples with the variational bound and ranks them.  The bound itself runs through the HIP kernels
(`ELBOMixin.elbo`: denoiser, k_elbo_rows, k_masked_sums); ranking is host logic over per-sample scalars and SMILES strings,
as in the reference.  SMILES conversion (rdkit) stays outside this package: `sort_by_elbo` takes the strings as inputs."""
from __future__ import annotations

import copy

import numpy as np
import torch

from . import graph
from .diffusion_abstract import DiscreteDenoisingDiffusion
from .placeholder import PlaceHolder


def reactions_sorted_with_weighted_prob(restructured_data, lambda_value):
    """utils/graph.py:709-736 on arrays: per product, weighted_prob = lambda * softmax(-elbo) + (1 - lambda) * count share, reactions in
    descending order of it (ties keep their input order, as the reference's stable sort does).  Returns new dicts; the input is left
    untouched.  The two normalisers are accumulated left to right (cumsum), which is the reference's summation order bit for bit."""
    ranked = {}
    for product, reactions in restructured_data.items():
        p_elbo = np.exp(-np.array([rx['elbo'] for rx in reactions], dtype=np.float64))
        counts = np.array([rx['count'] for rx in reactions], dtype=np.float64)
        w = p_elbo / np.cumsum(p_elbo)[-1] * lambda_value + counts / np.cumsum(counts)[-1] * (1 - lambda_value) if len(reactions) else p_elbo
        ranked[product] = [dict(copy.deepcopy(reactions[i]), weighted_prob=w[i]) for i in np.argsort(-w, kind='stable')]
    return ranked


def _to_device(batch, dev):
    """A PyG-style sparse batch (mapping or attribute object) with its tensors on `dev`."""
    items = batch.items() if isinstance(batch, dict) else vars(batch).items()
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in items}


class DiscreteDenoisingDiffusionRxn(DiscreteDenoisingDiffusion):
    """diffusion/diffusion_rxn.py: scoring and ranking of samples (the chemistry-side methods -- SMILES, plots, dataset
    evaluation -- are out of scope)."""

    @torch.no_grad()
    def sample_for_condition(self, dense_data, n_samples, inpaint_node_idx, inpaint_edge_idx, device=None, noise_fn=None,
                             chain_seeds=None):
        """diffusion_rxn.py:107-128: n_samples chains per condition of `dense_data` (duplicated IN PLACE, as the reference's
        graph.duplicate_data does: sample_n_conditions relies on it), inpainting lists repeated per sample; returns the final
        samples collapsed to class indices (X (bs*n_samples, n), E (bs*n_samples, n, n)), which is what
        sample_one_batch(get_true_rxns=True, get_chains=True) hands back.  noise_fn / chain_seeds: see sample_one_batch."""
        dup = graph.duplicate_data(dense_data=dense_data, n_samples=n_samples, get_discrete_data=False)
        rep = lambda idx: [item for item in idx for _ in range(n_samples)] if idx is not None else None
        final_samples = self.sample_one_batch(data=dup, inpaint_node_idx=rep(inpaint_node_idx), inpaint_edge_idx=rep(inpaint_edge_idx),
                                              get_true_rxns=True, get_chains=True, device=device, noise_fn=noise_fn,
                                              chain_seeds=chain_seeds, record_chains=False)[0]
        return final_samples

    @torch.no_grad()
    def sample_n_conditions(self, dataloader, inpaint_node_idx, inpaint_edge_idx, device_to_use=None, epoch_num=None, seed=0,
                            group=None, to_smiles=None):
        """diffusion_rxn.py:31-105 for one process per GPU: the first cfg.test.n_conditions conditions of `dataloader` (an
        iterable of PyG-style sparse batches -- x, edge_index, edge_attr, batch, y, mask_atom_mapping, mol_assignment -- or of
        dense PlaceHolders), cfg.test.n_samples_per_condition chains each, (condition, sample) units sharded over the ranks of
        `group` (distributed.sample_conditions_sharded; the reference walks the conditions in a Python loop, num_gpus per
        iteration under nn.DataParallel), dense conversion on the device (graph.to_dense).
        Returns, on rank 0, (all_gen_rxn_smiles, all_true_rxn_smiles, all_gen_rxn_dense, all_true_rxn_dense): per condition the
        generated / true reactions as collapsed CPU PlaceHolders of n_samples rows (the reference's .pyg() objects hold the
        same X / E indices) and, when `to_smiles(X, E) -> list[str]` is given (rdkit lives outside this package), their SMILES;
        (None,) * 4 on the other ranks."""
        assert epoch_num is not None, f'Need to provide epoch_num to use this function. Got epoch_num={epoch_num}.'
        from . import distributed
        dev = torch.device(device_to_use) if device_to_use is not None else self.device_
        n_cond, n_samples = int(self.cfg.test.n_conditions), int(self.cfg.test.n_samples_per_condition)
        conditions = []
        for batch in dataloader:
            dense = batch if isinstance(batch, PlaceHolder) else graph.to_dense(_to_device(batch, dev))
            dense = dense.to_device(dev)
            for k in range(dense.X.shape[0]):
                nk = int(dense.node_mask[k].sum().item())          # trim to the graph's own size, as a one-graph to_dense gives
                one = dense.subset_by_idx(k, k + 1)
                conditions.append(PlaceHolder(X=one.X[:, :nk].contiguous(), E=one.E[:, :nk, :nk].contiguous(), y=one.y,
                                              node_mask=one.node_mask[:, :nk].contiguous(),
                                              atom_map_numbers=one.atom_map_numbers[:, :nk].contiguous() if one.atom_map_numbers is not None else None,
                                              mol_assignments=one.mol_assignments[:, :nk].contiguous() if one.mol_assignments is not None else None))
                if len(conditions) == n_cond:
                    break
            if len(conditions) == n_cond:
                break
        assert inpaint_node_idx is None or len(inpaint_node_idx) == len(conditions), 'length of inpaint_node_idx and the number of conditions differ'
        assert inpaint_edge_idx is None or len(inpaint_edge_idx) == len(conditions), 'length of inpaint_edge_idx and the number of conditions differ'
        res = distributed.sample_conditions_sharded(conditions, n_samples, self.sample_for_condition, seed=seed,
                                                    inpaint_node_idx=inpaint_node_idx, inpaint_edge_idx=inpaint_edge_idx, group=group)
        if res is None:
            return None, None, None, None
        X, E = res[0].cpu(), res[1].cpu()
        gen_dense, true_dense, gen_smiles, true_smiles = [], [], [], []
        for c, cond in enumerate(conditions):
            nk = cond.X.shape[1]
            rows = slice(c * n_samples, (c + 1) * n_samples)
            gen = PlaceHolder(X=X[rows, :nk].long(), E=E[rows, :nk, :nk].long(), y=torch.zeros(n_samples, 0),
                              node_mask=cond.node_mask.cpu().expand(n_samples, -1).clone(),
                              atom_map_numbers=cond.atom_map_numbers.cpu().expand(n_samples, -1).clone() if cond.atom_map_numbers is not None else None,
                              mol_assignments=cond.mol_assignments.cpu().expand(n_samples, -1).clone() if cond.mol_assignments is not None else None)
            tc = cond.get_new_object().mask(collapse=True)
            true = gen.get_new_object(X=tc.X.cpu().expand(n_samples, -1), E=tc.E.cpu().expand(n_samples, -1, -1))
            gen_dense.append(gen); true_dense.append(true)
            if to_smiles is not None:
                gen_smiles.append(to_smiles(gen.X, gen.E)); true_smiles.append(to_smiles(true.X, true.E))
        return gen_smiles, true_smiles, gen_dense, true_dense

    @torch.no_grad()
    def estimate_elbo_with_repeats(self, final_samples_one_hot, noise_fn=None):
        """diffusion_rxn.py:606-641: mean over cfg.test.repeat_elbo estimates of the per-sample bound, the batch cut into chunks
        of cfg.test.elbo_batch_size samples.  Everything stays on the device (the reference reduces on the CPU)."""
        total = final_samples_one_hot.X.shape[0]
        step = getattr(self.cfg.test, 'elbo_batch_size', None)
        step = total if not step or step <= 0 else min(int(step), total)
        repeats = int(self.cfg.test.repeat_elbo)
        # running sums of the three per-sample terms (elbo, loss_t, loss_0) on the device; a chunk writes its own slice, so neither the
        # per-repeat lists nor the concatenations of the reference exist here.  mean over repeats = sum / repeats (fp32, repeat order)
        acc = torch.zeros(3, total, dtype=torch.float32, device=final_samples_one_hot.X.device)
        for _ in range(repeats):
            for lo in range(0, max(total, 1), max(step, 1)):
                hi = min(lo + step, total)
                part = final_samples_one_hot if (lo, hi) == (0, total) else final_samples_one_hot.subset_by_idx(lo, hi)
                terms = self.elbo(dense_true=part, avg_over_batch=False, noise_fn=noise_fn)
                for k, v in enumerate(terms):
                    acc[k, lo:hi] += v.float()
        acc /= repeats
        return acc[0], acc[1], acc[2]                                   # elbos, loss_t, loss_0: (bs * n_samples,)

    @torch.no_grad()
    def compare_true_and_gen_atoms(self, true_X, gen_X, n, device=None):
        """diffusion_rxn.py:538-566: 1.0 for every sample whose multiset of atom classes equals the true reaction's (both (N, n)
        class indices; compared after sorting each row)."""
        dev = true_X.device
        same = (true_X.sort(dim=-1)[0] == gen_X.to(dev).sort(dim=-1)[0]).sum(dim=-1)
        return (same == n).float()

    @torch.no_grad()
    def score_one_batch(self, final_samples, true_data, bs, n_samples, n, device=None, to_rxn_smiles=None, noise_fn=None):
        """diffusion_rxn.py:495-536 without the rdkit-bound validity scores: canonical strings of the generated and the true reactions
        (`to_rxn_smiles(X, E, atom_map_numbers) -> ['r1.r2>>p', ...]`, the caller's chemistry: mol.get_cano_smiles_from_dense in the
        reference; default: the object's `rxn_smiles_fn`), duplicates dropped per condition, the bound of every distinct sample
        estimated on the device (estimate_elbo_with_repeats), samples ranked by it and by the count-weighted probability
        (sort_by_elbo, reactions_sorted_with_weighted_prob), top-k accuracy and reciprocal rank against the true reaction
        (calculate_top_k), and the share of distinct samples with the right atoms per condition (compare_true_and_gen_atoms).
        final_samples / true_data: collapsed PlaceHolders (class indices) of bs * n_samples graphs, condition-major.
        Returns (scores, elbo_sorted_rxns, weighted_prob_sorted_rxns) as the reference does."""
        import torch.nn.functional as F
        to_rxn_smiles = to_rxn_smiles or getattr(self, 'rxn_smiles_fn', None)
        if to_rxn_smiles is None:
            raise NotImplementedError('score_one_batch needs to_rxn_smiles (dense classes -> canonical reaction strings): rdkit is not part '
                                      'of this repository')
        gen_rxn = list(to_rxn_smiles(final_samples.X, final_samples.E, getattr(final_samples, 'atom_map_numbers', None)))
        true_rxn = list(to_rxn_smiles(true_data.X, true_data.E, getattr(true_data, 'atom_map_numbers', None)))
        unique_indices, counts, is_unique = graph.get_unique_indices_from_reaction_list(gen_rxn)
        keep = torch.tensor(is_unique, device=final_samples.X.device)
        final_u, true_u = final_samples.select_subset(keep).get_new_object(), true_data.select_subset(keep).get_new_object()
        gen_u = [r for r, u in zip(gen_rxn, is_unique) if u]
        # share of a condition's distinct samples that have the true reaction's atoms (:406, 421-431)
        match = self.compare_true_and_gen_atoms(true_u.X, final_u.X, n).cpu()
        ui = torch.tensor(unique_indices)
        scores = {'matching_atoms': torch.stack([match[(ui >= i * n_samples) & (ui < (i + 1) * n_samples)].mean() for i in range(bs)])}
        kx, ke = len(self.dataset_info.atom_decoder), len(self.dataset_info.bond_decoder)
        elbos, loss_t, loss_0 = self.estimate_elbo_with_repeats(
            final_u.get_new_object(X=F.one_hot(final_u.X.long(), kx).float(), E=F.one_hot(final_u.E.long(), ke).float()), noise_fn=noise_fn)
        rcts, prods = graph.split_reactions_to_reactants_and_products(gen_u)
        counts = list(counts)
        elbos = elbos.clone()
        for i, rct in enumerate(rcts):                 # dummy samples '[U]>>product' parse to empty reactants: never ranked (:459-463)
            if all(a == '' for a in rct):
                elbos[i] = float('inf'); counts[i] = 0
        topks = list(getattr(self.cfg.test, 'topks', []) or [])
        by_elbo = self.sort_by_elbo(elbos, loss_t, loss_0, rcts, prods, is_unique=is_unique, counts=counts, bs=bs, n_samples=n_samples, k=topks)
        lam = self.cfg.test.sort_lambda_value
        by_weight = reactions_sorted_with_weighted_prob(by_elbo, lam)
        if topks:
            t_rcts, t_prods = graph.split_reactions_to_reactants_and_products(true_rxn)
            t_rcts, t_prods = t_rcts[::n_samples], t_prods[::n_samples]            # one entry per condition (:476-477)
            topk, mrr = graph.calculate_top_k(self.cfg, by_elbo, t_rcts, t_prods)
            topk_w, mrr_w = graph.calculate_top_k(self.cfg, by_weight, t_rcts, t_prods)
            scores['mrr'], scores[f'mrr_weighted_{lam}'] = mrr, mrr_w
            for j, k_ in enumerate(topks):
                scores[f'top-{k_}'], scores[f'top-{k_}_weighted_{lam}'] = topk[:, j], topk_w[:, j]
        return scores, by_elbo, by_weight

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
