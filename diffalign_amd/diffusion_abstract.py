"""DiscreteDenoisingDiffusion host class with the reference's constructor and method names
(diffalign/diffusion/diffusion_abstract.py:32-372).  Device is explicit (the reference's module-level `device`
global is gone); cfg is any attribute-style config (OmegaConf / SimpleNamespace) with the reference's keys."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import graph, helpers
from .noise_schedule import (AbsorbingStateTransitionMaskNoEdge, DiscreteUniformTransition, MarginalUniformTransition)
from .placeholder import PlaceHolder
from .elbo import ELBOMixin
from .sampling import SamplingMixin
from .training import TrainingMixin
from .transformer_model_with_y import GraphTransformerWithY, GraphTransformerWithYAtomMapPosEmb


class DiscreteDenoisingDiffusion(SamplingMixin, ELBOMixin, TrainingMixin, nn.Module):
    def __init__(self, cfg, dataset_infos, node_type_counts_unnormalized=None, edge_type_counts_unnormalized=None,
                 use_data_parallel=None, device='cuda:0'):
        super().__init__()
        if use_data_parallel:
            raise NotImplementedError('nn.DataParallel is replaced by one process per GPU (diffalign_amd.distributed)')
        input_dims = dict(dataset_infos.input_dims)
        output_dims = dict(dataset_infos.output_dims)
        if cfg.neuralnet.extra_features:
            input_dims = {'X': input_dims['X'] + 8, 'E': input_dims['E'], 'y': input_dims['y'] + 12}   # :38-40
        self.cfg = cfg
        self.T = cfg.diffusion.diffusion_steps
        self.Xdim_output, self.Edim_output = output_dims['X'], output_dims['E']
        if cfg.neuralnet.increase_y_dim_for_multigpu:
            output_dims['y'] += 1                                                                       # :48-49
        self.ydim_output = output_dims['y']
        self.node_dist = getattr(dataset_infos, 'nodes_dist', None)
        self.dataset_info = dataset_infos
        self.eps = 1e-6
        self.device_ = torch.device(device)
        self.extra_features_fn = None      # optional override: callable(z_t) -> (X_extra (B,n,8), y_extra (B,12)); see compute_extra_data
        self._extra_calc = None
        self.pos_encodings_fn = None       # optional override: callable(z_t) -> (B,n,dx); see get_pos_encodings
        self.pos_emb_module = None
        node_idx, edge_idx = graph.get_index_from_states(self.dataset_info.atom_decoder, self.dataset_info.bond_decoder,
                                                         list(cfg.diffusion.node_states_to_mask),
                                                         cfg.diffusion.edge_states_to_mask)
        abs_e, abs_x = 0, self.dataset_info.atom_decoder.index('Au')                                   # :67-68
        nn_ = cfg.neuralnet
        if nn_.architecture == 'with_y':
            self.model = GraphTransformerWithY(n_layers=nn_.n_layers, input_dims=input_dims, hidden_mlp_dims=nn_.hidden_mlp_dims,
                                               hidden_dims=nn_.hidden_dims, output_dims=output_dims, act_fn_in=nn.ReLU(),
                                               act_fn_out=nn.ReLU(), improved=nn_.improved, dropout=nn_.dropout)
        elif nn_.architecture == 'with_y_atommap_number_pos_enc':
            self.model = GraphTransformerWithYAtomMapPosEmb(
                n_layers=nn_.n_layers, input_dims=input_dims, hidden_mlp_dims=nn_.hidden_mlp_dims, hidden_dims=nn_.hidden_dims,
                output_dims=output_dims, act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), pos_emb_permutations=nn_.pos_emb_permutations,
                improved=nn_.improved, dropout=nn_.dropout, p_to_r_skip_connection=nn_.p_to_r_skip_connection,
                p_to_r_init=nn_.p_to_r_init, alignment_type=nn_.alignment_type, input_alignment=nn_.input_alignment)
        else:
            raise ValueError(f'Unknown architecture: {nn_.architecture}')
        self.ema = None
        if cfg.neuralnet.use_ema:                                                                       # :94-97
            from .ema import EMA
            self.ema = EMA(self.model, beta=cfg.neuralnet.ema_decay, power=1)
        steps = [cfg.diffusion.diffusion_steps, cfg.diffusion.diffusion_steps_eval]
        tr = cfg.diffusion.transition
        if tr == 'uniform':
            mk = lambda T_: DiscreteUniformTransition(cfg.diffusion.diffusion_noise_schedule, T_, self.Xdim_output, self.Edim_output,
                                                      self.ydim_output, cfg.diffusion.diffuse_edges, node_idx, edge_idx, device)
        elif tr == 'marginal':
            xm = self.dataset_info.node_types.float() / self.dataset_info.node_types.float().sum()
            em = self.dataset_info.edge_types.float() / self.dataset_info.edge_types.float().sum()
            mk = lambda T_: MarginalUniformTransition(xm, em, self.ydim_output, cfg.diffusion.diffusion_noise_schedule, T_,
                                                      cfg.diffusion.diffuse_edges, node_idx, edge_idx, device)
        elif tr == 'absorbing_masknoedge':
            mk = lambda T_: AbsorbingStateTransitionMaskNoEdge(T_, self.Xdim_output, self.Edim_output, self.ydim_output,
                                                               cfg.diffusion.diffuse_edges, abs_x, abs_e, node_idx, edge_idx, device)
        else:
            raise ValueError(f'Unknown transition: {tr}')
        self.transition_model, self.transition_model_eval = mk(steps[0]), mk(steps[1])
        self.limit_dist = self.transition_model.get_limit_dist()
        self.to(device)

    # ---- a17 -------------------------------------------------------------------------------------------------
    def apply_noise(self, dense_data, t_int, transition_model, noise=None):
        """diffusion_abstract.py:227-264."""
        assert dense_data.X.dim() == 3
        Qtb = transition_model.get_Qt_bar(t_int)
        pX, pE = helpers.noise_probs(dense_data, Qtb)
        prob_t = dense_data.get_new_object(X=pX, E=pE, y=t_int.float().to(pX.device))
        z_t = helpers.sample_discrete_features(prob=prob_t, noise=noise)
        return z_t, prob_t

    def get_pos_encodings(self, z_t):
        """diffusion_abstract.py:266-284: positional encodings by cfg.neuralnet.pos_encoding_type, computed on the device
        (diffalign_amd/pos_encodings.py).  `pos_encodings_fn` (callable(z_t) -> (B,n,dx)) overrides them."""
        if self.cfg.neuralnet.architecture != 'with_y_atommap_number_pos_enc':
            return None
        if self.pos_encodings_fn is not None:
            return self.pos_encodings_fn(z_t)
        if self.pos_emb_module is None:
            from .pos_encodings import PositionalEmbedding
            self.pos_emb_module = PositionalEmbedding(self.cfg.neuralnet.hidden_dims['dx'], self.cfg.neuralnet.pos_emb_permutations)
        kind, pe = self.cfg.neuralnet.pos_encoding_type, self.pos_emb_module
        if kind == 'smiles_pos_enc':
            return pe.matched_positional_encodings_sinusoidal(z_t.atom_map_numbers, z_t.mol_assignments, direction='retro')
        if kind in ('laplacian_pos_enc', 'laplacian_pos_enc_gpu'):
            return pe.matched_positional_encodings_laplacian(z_t.E.argmax(-1), z_t.atom_map_numbers, z_t.mol_assignments,
                                                             self.cfg.neuralnet.num_lap_eig_vectors, direction='retro')
        if kind == 'infoleak_pos_enc':
            return pe.matched_positional_encodings_infoleak(z_t.atom_map_numbers, z_t.mol_assignments, direction='retro')
        if kind == 'gaussian_pos_enc':
            return pe.matched_positional_encodings_gaussian(z_t.atom_map_numbers, z_t.mol_assignments)
        raise ValueError(f'pos_encoding_type {kind} not recognized')

    def get_pos_encodings_if_relevant(self, z_t):
        return self.get_pos_encodings(z_t)

    def compute_extra_data(self, noisy_data, pad_noise=None, step=None):
        """diffusion_abstract.py:350-372: concatenate the 8 node / 12 global extra features (ExtraFeatures('all') +
        ExtraMolecularFeatures).  Computed on the device (diffalign_amd/extra_features.py); the reference
        moves E, X and node_mask to the CPU for this at every step.  `extra_features_fn` (callable(z_t) -> (X_extra, y_extra))
        overrides the built-in calculators, e.g. to feed precomputed features.
        `step`: the fused reverse step's context: the join with the eigensolver's stream is then left to the engine
        (step['inputs_ready']), which enqueues the edge input stage in front of it."""
        if self.extra_features_fn is not None:
            X_extra, y_extra = self.extra_features_fn(noisy_data)
            Xw, yw = torch.cat((noisy_data.X, X_extra), -1), torch.cat((noisy_data.y, y_extra), -1)
        else:
            from . import extra_features as XF
            if self._extra_calc is None:
                self._extra_calc = (XF.ExtraFeatures('all', self.dataset_info), XF.ExtraMolecularFeatures(self.dataset_info))
            # `eig_cache`: set on the chain's state object by sample_one_batch (the fixed product block's part of the eigensolver
            # is kept along the chain); objects derived from it (get_new_object) do not carry it
            if pad_noise is None:
                pad_noise = getattr(noisy_data, 'pad_noise', None)                # per chain (sample_one_batch)
            # base=: the widened rows are assembled in place, nothing is left to concatenate behind the eigensolver
            Xw, yw, pending = XF.compute_extra_features(noisy_data.X, noisy_data.E, noisy_data.node_mask, *self._extra_calc, pad_noise=pad_noise,
                                                        eig_cache=getattr(noisy_data, 'eig_cache', None), base=(noisy_data.X, noisy_data.y),
                                                        defer_join=True)
            if pending is not None:
                if step is not None and os.environ.get('DA_DEFER_FEATURE_JOIN', '1') != '0':      # (read per call: an A/B switch)
                    step['inputs_ready'] = pending
                else:
                    torch.cuda.current_stream().wait_event(pending[0])
        # a view of noisy_data with widened X and y: the reference's get_new_object clones every field (E: 161 MB at the
        # benchmark size, 60 us per step) for an object that only feeds the denoiser
        return PlaceHolder(X=Xw, E=noisy_data.E, y=yw, node_mask=noisy_data.node_mask, atom_map_numbers=noisy_data.atom_map_numbers,
                           mol_assignments=noisy_data.mol_assignments)

    def reverse_step(self, z_t, t_array, pos_encodings=None, transition_model=None, temperature=(1.0, 1.0), noise=None, chain_seeds=None,
                     call=0, orig=None, node_masks=None, generator=None, want_logits=False, t_int=None, tables=None, out=None, out_holds_condition=False, last_rows=None):
        """One reverse step p(z_s | z_t) (diffusion/sampling.py:76-86,165-178): denoiser forward + posterior + categorical draw +
        apply_mask.  With 7 edge classes the edge half is drawn inside the denoiser's output kernel (da_edge_out_sample; the
        (B,n,n,7) logits and noise tensors never reach HBM); otherwise forward() + helpers.posterior_sample.  Both give the same
        z_s for the same noise, bit for bit.  Returns (zs_X, zs_E, pred); pred.E is None in the fused form (want_logits=True: separate form).
        `t_int`: the step number when all graphs share it (then t_array is not read for the tables).
        `out` = (X, E): preallocated fp32 tensors that receive z_s (and are returned); they must not be z_t's.
        `out_holds_condition`: the caller's promise that out's E already has the condition's values (`orig`) at every pair that is not
        free to diffuse -- true for a buffer that held an earlier state of the same chain: those pairs are then not written again.
        `last_rows`: graph.last_layers_plan(node_masks) of this chain (computed once per chain: host syncs), or graph.free_rows' list
        alone: the denoiser's last layer then computes the query rows of free nodes only -- nothing else of it is read by the fused
        step (pred.X is then defined on those rows only) --, and with a plan the last two layers form the edge features of the pairs
        somebody reads only (graph.LastLayersPlan)."""
        from . import helpers
        tm = transition_model if transition_model is not None else self.transition_model_eval
        if tables is not None:       # (Q_t, Qbar_{t-1}, Qbar_t) of this step laid out by the caller (graphs.GraphedChain)
            Qt, Qsb, Qtb = tables
        elif t_int is not None:      # every graph is at step t_int (the sampling loop): slices of tables laid out once per batch size
            Qt, Qsb, Qtb = tm.step_tables(int(t_int), z_t.X.shape[0])
        else:
            Qt, Qsb, Qtb = tm.get_Qt(t_array), tm.get_Qt_bar(t_array - 1), tm.get_Qt_bar(t_array)
        if self._net().supports_fused_step() and not self._net()._differentiable() and not want_logits:
            ctx = helpers.reverse_step_ctx(z_t, Qt, Qsb, Qtb, temperature, noise, chain_seeds, call, orig, node_masks, generator, out=out,
                                           out_holds_condition=out_holds_condition, last_rows=last_rows)
            pred = self.forward(z_t=z_t, pos_encodings=pos_encodings, step=ctx)
            return ctx['zs_X'], ctx['zs_E'], pred
        B, n, kx = z_t.X.shape
        if noise is None and chain_seeds is not None:
            noise = helpers.counter_exp_noise(chain_seeds.to(z_t.X.device), call, B, n, kx, z_t.E.shape[-1])
        pred = self.forward(z_t=z_t, pos_encodings=pos_encodings)
        _, _, zsX, zsE = helpers.posterior_sample(pred.X, pred.E, z_t, Qt, Qsb, Qtb, temperature=temperature, noise=noise, orig=orig,
                                                  node_masks=node_masks, generator=generator)
        if out is not None:
            out[0].copy_(zsX); out[1].copy_(zsE)
            zsX, zsE = out
        return zsX, zsE, pred

    def _net(self):
        """The denoiser forward() runs: the EMA copy outside training when cfg.neuralnet.use_ema (:333-339), else the model."""
        return self.ema.ema_model if (self.ema is not None and not self.training) else self.model

    def forward(self, z_t, use_pos_encoding_if_applicable=None, pos_encodings=None, step=None):
        """diffusion_abstract.py:293-348 (denoiser='neuralnet').  The reference's random relabelling of atom-map
        numbers (:297-300) is output-invariant and is skipped (it only consumes RNG and a host sync).
        `step`: helpers.reverse_step_ctx(...) -- the reverse step fused behind the output stage (see reverse_step)."""
        if self.cfg.diffusion.denoiser == 'random-uniform':                # :312-317: equal node logits, E as it came in
            return z_t.get_new_object(X=torch.ones_like(z_t.X))
        if self.cfg.diffusion.denoiser != 'neuralnet':
            # 'carbon' (:304-311) reads mol.atom_types / mol.atom_type_offset, which utils/mol.py does not define: it cannot run there either
            raise NotImplementedError(f"denoiser={self.cfg.diffusion.denoiser!r}: 'neuralnet' and 'random-uniform' are built")
        z_in = self.compute_extra_data(z_t, step=step) if self.cfg.neuralnet.extra_features else z_t
        if self.cfg.neuralnet.architecture == 'with_y_atommap_number_pos_enc':
            assert z_t.mol_assignments is not None, 'molecule_assigments is None in forward()'
            if pos_encodings is None:
                pos_encodings = self.get_pos_encodings(z_t)
            if use_pos_encoding_if_applicable is not None:
                pos_encodings = pos_encodings * use_pos_encoding_if_applicable[:, None, None].float()
            X, E, y, node_mask = self._net()(z_in.X, z_in.E, z_in.y, z_in.node_mask, z_in.atom_map_numbers, pos_encodings,
                                             z_in.mol_assignments, **({'step': step} if step is not None else {}))
        else:
            X, E, y, node_mask = self._net()(z_in.X, z_in.E, z_in.y, z_in.node_mask, **({'step': step} if step is not None else {}))
        return PlaceHolder(X=X, E=E, y=y, node_mask=node_mask, atom_map_numbers=z_t.atom_map_numbers,
                           mol_assignments=z_t.mol_assignments)
