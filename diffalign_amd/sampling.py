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
                         inpaint_node_idx=None, inpaint_edge_idx=None, noise_fn=None):
        """noise_fn(call_index, shape) -> Exp(1) tensor makes the draws a function of explicit noise (tests); the
        call order equals the reference's multinomial call order (X then E; limit draw first)."""
        assert data is not None, 'diffalign_amd samples conditionally: `data` (the dense condition) is required'
        if self.cfg.diffusion.classifier_free_guidance:
            raise NotImplementedError('classifier-free guidance is off in both align_* experiments and is not built')
        assert self.cfg.test.with_denoising
        dense_data = data
        device = dense_data.X.device if device is None else device
        node_mask = dense_data.node_mask
        B, n = node_mask.shape
        kx, ke = dense_data.X.shape[-1], dense_data.E.shape[-1]
        calls = [0]

        def nz():
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
        inpaint = inpaint_node_idx is not None or inpaint_edge_idx is not None
        if inpaint:
            fix_X, fix_E = graph.fix_nodes_and_edges_by_idx(dense_data, inpaint_node_idx, inpaint_edge_idx)

        def fix(z):
            if inpaint:
                z.X = torch.where(fix_X[..., None], dense_data.X, z.X)
                z.E = torch.where(fix_E[..., None], dense_data.E, z.E)
            if not self.cfg.diffusion.diffuse_edges: z.E = dense_data.E.clone()
            if not self.cfg.diffusion.diffuse_nodes: z.X = dense_data.X.clone()
            return z

        z_t = fix(z_t)
        T_eval = self.cfg.diffusion.diffusion_steps_eval
        assert self.T % T_eval == 0, 'diffusion_steps_eval should be divisible by diffusion_steps'
        eval_step_size = self.T // T_eval
        sample_chains = []
        temps = (float(self.cfg.diffusion.temperature_scaling_node), float(self.cfg.diffusion.temperature_scaling_edge))
        tm = self.transition_model_eval
        for t_int in range(T_eval, 0, -1):
            t_array = torch.full((B, 1), t_int, dtype=torch.long, device=device)
            z_t.y = t_array.float() * eval_step_size                       # raw train-scale step index (sampling.py:76)
            pred = self.forward(z_t=z_t, pos_encodings=pos_encodings)
            if get_chains:
                sample_chains.append((t_int, z_t.get_new_object().mask(z_t.node_mask, collapse=True)))
            _, _, zsX, zsE = helpers.posterior_sample(pred.X, pred.E, z_t, tm.get_Qt(t_array), tm.get_Qt_bar(t_array - 1),
                                                      tm.get_Qt_bar(t_array), temperature=temps, noise=nz(),
                                                      orig=dense_data, node_masks=masks)
            z_t.X, z_t.E = zsX, zsE
            z_t = fix(z_t)
        if get_chains:
            sample_chains.append((0, z_t.get_new_object().mask(z_t.node_mask, collapse=True)))
        if get_true_rxns:
            return (z_t.mask(z_t.node_mask, collapse=True), sample_chains, [], [], dense_data)
        if get_chains:
            return (z_t.mask(z_t.node_mask, collapse=True), sample_chains, [], [])
        return z_t
