"""SamplingMixin with the reference's sample_one_batch signature (diffalign/diffusion/sampling.py:11-197).

Per reverse step the reference runs forward -> log_softmax*temperature -> get_p_zs_given_zt -> sample_discrete_features
-> assert symmetric (device sync) -> apply_mask (builds a (B*n)^2 outer product) -> fix_nodes_and_edges_by_idx
(deepcopy of the batch).  Here a step is: denoiser forward (HIP) + ONE fused posterior/draw/apply_mask kernel; the
free-to-diffuse masks are computed once per batch and no host sync happens inside the loop."""
from __future__ import annotations

import torch

from . import graph, helpers


class SamplingMixin:
    @torch.no_grad()
    def sample_one_batch(self, device=None, n_samples=None, data=None, get_chains=False, get_true_rxns=False,
                         inpaint_node_idx=None, inpaint_edge_idx=None, noise_fn=None, chain_seeds=None, record_chains=None):
        """noise_fn(call_index, shape) -> Exp(1) tensor makes the draws a function of explicit noise (tests); the
        call order equals the reference's multinomial call order (X then E; limit draw first).
        chain_seeds (B,) int64: per-chain counter-based noise (helpers.counter_exp_noise) -- every chain's draws depend on its
        own seed only, so sharded runs give the same samples under any rank layout (diffalign_amd/distributed.py).
        record_chains=False: keep the return structure of get_chains=True but record no intermediate states (sample_for_condition reads
        element 0 only); then no step leaves the fused path, and small batches replay their steps as HIP graphs (graphs.GraphedChain)."""
        assert data is not None, 'diffalign_amd samples conditionally: `data` (the dense condition) is required'
        assert self.cfg.test.with_denoising
        dense_data = data
        device = dense_data.X.device if device is None else device
        node_mask = dense_data.node_mask
        B, n = node_mask.shape
        kx, ke = dense_data.X.shape[-1], dense_data.E.shape[-1]
        calls = [0]

        if chain_seeds is None and noise_fn is None:
            # no explicit noise: one seed per chain from torch's generator, once; every draw of the chain is then counter noise of
            # (seed, call index) -- the same distribution as fresh seeds per step without two generator kernels in front of each step
            chain_seeds = torch.randint(0, 2 ** 62, (B,), device=device)
        if chain_seeds is not None:
            assert noise_fn is None and chain_seeds.shape == (B,)
            chain_seeds = chain_seeds.to(device=device, dtype=torch.int64)

        def nz(step=False):
            """Noise of the next draw.  step=True (the reverse steps): keyword arguments for reverse_step -- with chain seeds the
            (B,n,n,ke) noise tensor is never materialised (the fused kernel generates the same values in registers)."""
            if chain_seeds is not None:
                calls[0] += 1
                if step:
                    return dict(chain_seeds=chain_seeds, call=calls[0] - 1)
                return helpers.counter_exp_noise(chain_seeds, calls[0] - 1, B, n, kx, ke)
            if step:
                return dict(noise=nz())
            if noise_fn is None:
                return None
            qx = noise_fn(calls[0], (B * n, kx)).reshape(B, n, kx).to(device)
            qe = noise_fn(calls[0] + 1, (B * n * n, ke)).reshape(B, n, n, ke).to(device)
            calls[0] += 2
            return qx, qe

        pos_encodings = self.get_pos_encodings_if_relevant(dense_data)
        z_t = helpers.sample_from_noise(limit_dist=self.limit_dist, node_mask=node_mask, T=self.T, noise=nz())
        z_t = dense_data.get_new_object(X=z_t.X, E=z_t.E, y=z_t.y)
        masks = graph.get_node_masks(dense_data, self.dataset_info.atom_decoder, self.dataset_info.bond_decoder,
                                     self.cfg.diffusion.mask_nodes, self.cfg.diffusion.mask_edges)
        z_t = graph.apply_mask(orig=dense_data, z_t=z_t, atom_decoder=self.dataset_info.atom_decoder,
                               bond_decoder=self.dataset_info.bond_decoder, node_masks=masks)
        # inpainting (sampling.py:42-45,175-178): the index lists go to the device once; per step the listed rows / pairs are
        # copied back from the condition by one scatter (the reference rebuilds dense masks from a deepcopy at every step)
        fix_nodes, fix_edges = graph.inpaint_lists(inpaint_node_idx, inpaint_edge_idx, device)

        def fix(z):
            graph.apply_inpainting(z, dense_data, fix_nodes, fix_edges)
            if not self.cfg.diffusion.diffuse_edges: z.E = dense_data.E.clone()
            if not self.cfg.diffusion.diffuse_nodes: z.X = dense_data.X.clone()
            return z

        z_t = fix(z_t)
        # the condition's side of the graph keeps its edges at every step: its share of the extra features' eigensolver (the
        # larger share: with absorbing noise the free side is mostly isolated atoms) is done once per chain (extra_features.EigCache)
        eig_cache = None
        if self.cfg.neuralnet.extra_features and self.extra_features_fn is None and self.cfg.diffusion.denoiser == 'neuralnet':
            from .extra_features import EigCache
            eig_cache = EigCache.from_free_masks(masks[1], node_mask)
        # noise on the padded slots of one eigenvector feature (extra_features.py:189: it only keeps 0 from becoming the mode there): drawn
        # once per chain instead of once per step
        pad_noise = torch.randn(B, n, device=device) if eig_cache is not None else None
        T_eval = self.cfg.diffusion.diffusion_steps_eval
        assert self.T % T_eval == 0, 'diffusion_steps_eval should be divisible by diffusion_steps'
        eval_step_size = self.T // T_eval
        sample_chains, prob_s_chains, pred_0_chains = [], [], []
        temps = (float(self.cfg.diffusion.temperature_scaling_node), float(self.cfg.diffusion.temperature_scaling_edge))
        tm = self.transition_model_eval
        guided = bool(getattr(self.cfg.diffusion, 'classifier_free_guidance', False))
        if guided:                                                         # what the unconditional forward may see (sampling.py:90-95)
            cond_free = graph.get_node_masks(dense_data, self.dataset_info.atom_decoder, self.dataset_info.bond_decoder,
                                             self.cfg.diffusion.mask_nodes, self.cfg.diffusion.mask_edges, include_supernode=False)
        log_every_t = int(getattr(getattr(self.cfg, 'train', None), 'log_every_t', 1) or 1)
        # step numbers and the y input of every step, laid out once (a fill, a cast and a multiply per step otherwise)
        all_t = torch.arange(T_eval + 1, device=device)[:, None, None].expand(-1, B, 1).contiguous()
        all_y = all_t.float() * eval_step_size
        recording = get_chains if record_chains is None else (get_chains and record_chains)
        from . import graphs
        if (not recording and not guided and chain_seeds is not None and self._net().supports_fused_step() and not self._net()._differentiable()
                and graphs.chain_graphs_wanted(B, n)):
            gc = graphs.GraphedChain.get(self, dense_data, masks, pos_encodings, temps, fix_nodes, fix_edges, eig_cache is not None, calls[0])
            gc.load(dense_data, masks, pos_encodings, fix_nodes, fix_edges, chain_seeds, z_t, eig_cache, pad_noise)
            zg = gc.run()
            z_t.X, z_t.E = zg.X.clone(), zg.E.clone()
            T_loop = range(0)
        else:
            T_loop = range(T_eval, 0, -1)
        holders = last_rows = None
        for t_int in T_loop:
            t_array = all_t[t_int]
            z_t.eig_cache, z_t.pad_noise = eig_cache, pad_noise            # read by compute_extra_data; not carried by get_new_object
            z_t.y = all_y[t_int]                                           # raw train-scale step index (sampling.py:76)
            s_int = t_int - 1
            record = recording and (s_int % log_every_t == 0 or s_int == self.T - 1)
            if not record and not guided:                                  # the whole step as one call (fused where the model allows)
                # the state alternates between two holders that start as the condition: the pairs that are not free to diffuse have
                # the condition's value at every step (apply_mask), so the fused step writes the free pairs only
                if holders is None:
                    holders = [(torch.empty(B, n, dense_data.X.shape[-1], device=device), dense_data.E.float().clone()) for _ in range(2)]
                out = holders[1] if z_t.E.data_ptr() == holders[0][1].data_ptr() else holders[0]
                if last_rows is None:                                      # (one host sync per chain)
                    last_rows = graph.last_layers_plan(masks)
                zsX, zsE, _ = self.reverse_step(z_t, t_array, pos_encodings=pos_encodings, transition_model=tm, temperature=temps,
                                                orig=dense_data, node_masks=masks, t_int=t_int, out=out, out_holds_condition=True,
                                                last_rows=last_rows, **nz(step=True))
                z_t.X, z_t.E = zsX, zsE
                z_t = fix(z_t)
                continue
            pred = self.forward(z_t=z_t, pos_encodings=pos_encodings)
            if record or guided:
                pX, pE, _, _ = helpers.posterior_sample(pred.X, pred.E, z_t, tm.get_Qt(t_array), tm.get_Qt_bar(t_array - 1),
                                                        tm.get_Qt_bar(t_array), temperature=temps, want_prob=True, want_sample=False)
            if guided:
                pX, pE = self._guide(z_t, t_array, pX, pE, cond_free)
            if record:                                                                   # sampling.py:120-161 (plotting chains)
                sample_chains.append((t_int, z_t.get_new_object().mask(z_t.node_mask, collapse=True)))
                # the reference softmaxes the temperature-scaled log-probabilities (sampling.py:82, 122-123), not the raw logits
                p0 = pred.get_new_object(X=torch.softmax(temps[0] * torch.log_softmax(pred.X, -1), -1),
                                         E=torch.softmax(temps[1] * torch.log_softmax(pred.E, -1), -1))
                p0.X[..., -1] = 0.                                         # SuNo predictions ignored (sampling.py:125-126)
                p0.X /= p0.X.sum(-1, keepdim=True)
                for chain, obj in ((pred_0_chains, p0), (prob_s_chains, z_t.get_new_object(X=pX, E=pE))):
                    obj = graph.apply_mask(orig=dense_data, z_t=obj, atom_decoder=self.dataset_info.atom_decoder,
                                           bond_decoder=self.dataset_info.bond_decoder, node_masks=masks)
                    chain.append((s_int, fix(obj)))
            if guided:                                                     # draw from the guided p(z_s | z_t) (sampling.py:165-174)
                z_s = helpers.sample_discrete_features(z_t.get_new_object(X=pX, E=pE), noise=nz())
                z_s = graph.apply_mask(orig=dense_data, z_t=z_s, atom_decoder=self.dataset_info.atom_decoder,
                                       bond_decoder=self.dataset_info.bond_decoder, node_masks=masks)
                z_t.X, z_t.E = z_s.X, z_s.E
                z_t = fix(z_t)
                continue
            _, _, zsX, zsE = helpers.posterior_sample(pred.X, pred.E, z_t, tm.get_Qt(t_array), tm.get_Qt_bar(t_array - 1),
                                                      tm.get_Qt_bar(t_array), temperature=temps, noise=nz(),
                                                      orig=dense_data, node_masks=masks)
            z_t.X, z_t.E = zsX, zsE
            z_t = fix(z_t)
        if get_chains:
            sample_chains.append((0, z_t.get_new_object().mask(z_t.node_mask, collapse=True)))
        if get_true_rxns:
            return (z_t.mask(z_t.node_mask, collapse=True), sample_chains, prob_s_chains, pred_0_chains, dense_data)
        if get_chains:
            return (z_t.mask(z_t.node_mask, collapse=True), sample_chains, prob_s_chains, pred_0_chains)
        return z_t

    def _guide(self, z_t, t_array, pX, pE, cond_free):
        """Classifier-free guidance (sampling.py:88-113): a second forward on z_t with the condition zeroed (everything apply_mask keeps
        fixed except the supernode; with classifier_free_full_unconditioning also the atom-map numbers) and the positional encodings
        switched off gives p_u(z_s | z_t); the step is drawn from softmax((1 + w) log p - w log p_u).  As in the reference the
        unconditional posterior uses the TRAINING transition model and no temperature."""
        d = self.cfg.diffusion
        B = z_t.X.shape[0]
        free_x, free_e = cond_free
        pair = free_e[:, :, None] & free_e[:, None, :]
        am = z_t.atom_map_numbers
        if d.classifier_free_full_unconditioning and self.cfg.neuralnet.architecture == 'with_y_atommap_number_pos_enc':
            am = torch.zeros_like(am)                                                                         # :100-101
        z_nc = z_t.get_new_object(X=z_t.X * free_x[:, :, None].float(), E=z_t.E * pair[..., None].float(), atom_map_numbers=am)
        pred_u = self.forward(z_t=z_nc, use_pos_encoding_if_applicable=torch.zeros(B, dtype=torch.bool, device=z_t.X.device))   # :105
        tm = self.transition_model
        uX, uE, _, _ = helpers.posterior_sample(pred_u.X, pred_u.E, z_t, tm.get_Qt(t_array), tm.get_Qt_bar(t_array - 1),
                                                tm.get_Qt_bar(t_array), want_prob=True, want_sample=False)        # :106-107
        w = float(d.classifier_free_guidance_weight)
        eps = 1e-30
        gX = torch.softmax((1 + w) * torch.log(pX + eps) - w * torch.log(uX + eps), -1)                      # :110-111
        gE = torch.softmax((1 + w) * torch.log(pE + eps) - w * torch.log(uE + eps), -1)
        gX[gX < 1e-28] = 0.                                                                                   # :112-113
        gE[gE < 1e-28] = 0.
        return gX, gE
