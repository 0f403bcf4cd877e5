"""Training step (diffalign/diffusion/diffusion_abstract.py:134-225, training/helpers.py:139-142).

`model.train()` + autograd on: `training_step` runs apply_noise -> apply_mask -> the differentiable fp32 denoiser forward
(train_forward.py) -> the objective ('ce' / 'vb' / 'vbce') as an autograd.Function, and returns a loss that `backward()`s into
every parameter; `optim.FlatAdamW` (fused AdamW with amsgrad over one flat buffer) and `distributed.allreduce_flat_gradient`
complete the step (`train_one_step`).  Loss values and gradient norms are pinned to the reference's own `loss.backward()`
(tests/golden/chain_elbo_tiny.npz), every parameter's gradient to autograd on the oracle.
`model.eval()` or no_grad: the same call evaluates the loss VALUES through the fused bf16 inference path (no graph).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib as L, elbo as E_, graph
from .placeholder import PlaceHolder


def ce_grad(pred: PlaceHolder, dense_true: PlaceHolder, node_masks, lambda_E: float = 5.0, diffuse_nodes=True, diffuse_edges=True):
    """d helpers.ce / d pred (helpers.py:153-191): what `loss.backward()` leaves in pred.X.grad / pred.E.grad."""
    X, E = dense_true.X.contiguous().float(), dense_true.E.contiguous().float()
    lX, lE = pred.X.contiguous().float(), pred.E.contiguous().float()
    B, n, kx = X.shape
    ke = E.shape[-1]
    dev = X.device
    mx, me = L.as_u8(node_masks[0]), L.as_u8(node_masks[1])
    rows0X, rows0E = torch.zeros(B, n, device=dev), torch.zeros(B, n, n, device=dev)
    sums = E_.masked_sums(rows0X, rows0E, node_masks)            # only the counts are used
    gX, gE = torch.empty_like(lX), torch.empty_like(lE)
    L.check(L.lib().da_ce_grad(L.ptr(X), L.ptr(E), L.ptr(lX), L.ptr(lE), L.ptr(mx), L.ptr(me), L.ptr(sums),
                               C.c_float(1.0 if diffuse_nodes else 0.0), C.c_float(float(lambda_E) if diffuse_edges else 0.0),
                               L.ptr(gX), L.ptr(gE), C.c_int(B), C.c_int(n), C.c_int(kx), C.c_int(ke), L.stream()), 'da_ce_grad')
    return gX, gE


class TrainingMixin:
    """Needs from the host object: cfg, T, transition_model, dataset_info, model, forward, apply_noise, elbo_batch_quick."""

    def training_step(self, data: PlaceHolder, i: int = 0, device=None, t_int: Optional[torch.Tensor] = None, noise=None,
                      z_t: Optional[PlaceHolder] = None, pos_encodings=None, want_terms: bool = True,
                      uncond_draw: Optional[torch.Tensor] = None):
        """diffusion_abstract.py:134-225.  `data`: dense PlaceHolder on the device (the reference's graph.to_dense is the data
        loader's side).  Returns (loss_X, loss_E, loss) as the reference does.  t_int / noise / z_t fix the reference's random draws;
        uncond_draw (B,) in [0,1) stands for the `torch.rand(bs)` of classifier-free guidance (:163)."""
        d = self.cfg.diffusion
        dense_data = data
        B = dense_data.X.shape[0]
        dev = dense_data.X.device
        if t_int is None:
            t_int = torch.randint(1, self.T + 1, size=(B, 1), device=dev)                                   # :136
        node_masks = E_._node_masks(self, dense_data)
        if z_t is None:
            z_t, _ = self.apply_noise(dense_data, t_int=t_int, transition_model=self.transition_model, noise=noise)   # :137
            z_t = graph.apply_mask(orig=dense_data, z_t=z_t, atom_decoder=self.dataset_info.atom_decoder,
                                   bond_decoder=self.dataset_info.bond_decoder, mask_nodes=d.mask_nodes,
                                   mask_edges=d.mask_edges, node_masks=node_masks)                            # :139-141
        z_keep = z_t.get_new_object()                                                                         # :144
        if not d.diffuse_edges: z_t.E = dense_data.E.clone()
        if not d.diffuse_nodes: z_t.X = dense_data.X.clone()
        use_pos = None
        if getattr(d, 'classifier_free_guidance', False):
            z_t, use_pos = self._drop_condition(dense_data, z_t, uncond_draw)                                 # :152-179
        loss_kind = getattr(getattr(self.cfg, 'train', None), 'loss', 'ce')
        if self.model.training and torch.is_grad_enabled():
            # the differentiable path: fp32 training forward under autograd (train_forward.py), the objective as one
            # autograd.Function whose backward is da_ce_grad / da_vb_grad; `loss.backward()` then fills every parameter's .grad
            from .autograd_ops import DiffusionLoss
            pred = self.forward(z_t=z_t, pos_encodings=pos_encodings,   # pos_encodings: precomputed (they depend on the condition only)
                                use_pos_encoding_if_applicable=use_pos)
            loss = DiffusionLoss.apply(pred.X, pred.E, self, dense_data, z_keep, node_masks, loss_kind)
            loss_X = loss_E = torch.zeros((1,))
            if loss_kind == 'ce' and want_terms:      # the per-element CE terms the reference also returns (a boolean gather: a host sync)
                with torch.no_grad():
                    rX, rE = E_.elbo_rows(E_.LOGP, dense_data, logits=pred.get_new_object(X=pred.X.detach(), E=pred.E.detach()))
                    mx, me = node_masks[0].bool(), node_masks[1].bool()
                    loss_X, loss_E = -rX[mx], -rE[me[:, :, None] & me[:, None, :]]
            return loss_X, loss_E, loss
        with torch.no_grad():
            pred = self.forward(z_t=z_t, use_pos_encoding_if_applicable=use_pos)                              # :181-185
        lam = d.lambda_train[0]
        loss_X = loss_E = torch.zeros((1,))
        if loss_kind in ('ce', 'vbce'):
            rX, rE = E_.elbo_rows(E_.LOGP, dense_data, logits=pred)
            s = E_.masked_sums(rX, rE, node_masks).double()
            ce = (-(s[:, 0].sum() / s[:, 1].sum()) * float(bool(d.diffuse_nodes))
                  - float(bool(d.diffuse_edges)) * lam * (s[:, 2].sum() / s[:, 3].sum())).float()          # helpers.py:186-189
        if loss_kind == 'ce':
            mx, me = node_masks[0].bool(), node_masks[1].bool()
            loss_X, loss_E, loss = -rX[mx], -rE[me[:, :, None] & me[:, None, :]], ce
        elif loss_kind == 'vb':
            loss = self.elbo_batch_quick(dense_data, pred=pred, z_t=z_keep, lambda_E=lam)                     # :212
        elif loss_kind == 'vbce':
            loss = self.elbo_batch_quick(dense_data, pred=pred, z_t=z_keep, lambda_E=lam) + d.ce_lambda * ce  # :221-222
        else:
            raise ValueError(f'Unknown loss {loss_kind}')
        return loss_X, loss_E, loss

    def _drop_condition(self, dense_data: PlaceHolder, z_t: PlaceHolder, uncond_draw: Optional[torch.Tensor]):
        """Classifier-free guidance at training time (diffusion_abstract.py:152-179): graphs whose draw falls below
        classifier_free_uncond_prob lose their condition -- everything apply_mask would keep fixed except the supernode is
        zeroed in z_t and their positional encodings are switched off.  classifier_free_drop_within_batch=False is the older rule:
        the whole batch follows graph 0's draw.  With classifier_free_full_unconditioning the reference clears the atom-map
        numbers of the graphs that KEEP their condition (:178, its own TODO); that is reproduced as written.
        Returns (z_t, use_pos_encoding_if_applicable)."""
        d = self.cfg.diffusion
        B = z_t.X.shape[0]
        dev = z_t.X.device
        free_x, free_e = graph.get_node_masks(dense_data, self.dataset_info.atom_decoder, self.dataset_info.bond_decoder, d.mask_nodes,
                                              d.mask_edges, include_supernode=False)                         # :155-160
        r = torch.rand(B) if uncond_draw is None else uncond_draw                                             # :163
        unconditional = (r.to(dev) < d.classifier_free_uncond_prob)                                           # :164
        if d.classifier_free_drop_within_batch:
            keep_all = ~unconditional                                                                         # :167-169
        else:
            keep_all = (~unconditional[0:1]).expand(B)                                                        # :171-173
        free_x = free_x | keep_all[:, None]
        free_e = free_e | keep_all[:, None]
        am = z_t.atom_map_numbers
        if d.classifier_free_full_unconditioning and self.cfg.neuralnet.architecture == 'with_y_atommap_number_pos_enc':
            am = am * unconditional[:, None].to(am.dtype)                                                     # :176-178
        pair = free_e[:, :, None] & free_e[:, None, :]
        z_t = z_t.get_new_object(X=z_t.X * free_x[:, :, None].float(), E=z_t.E * pair[..., None].float(), atom_map_numbers=am)  # :179
        return z_t, ~unconditional

    def train_one_step(self, data: PlaceHolder, optimizer, i: int = 0, t_int=None, noise=None, z_t=None, group=None, pos_encodings=None,
                       update_ema: bool = True):
        """training/helpers.py:130-142 for one batch: zero_grad, training_step, backward, gradient all-reduce over `group` (data
        parallel: one process per GPU, RCCL), fused AdamW step.  Returns the (detached) loss."""
        from . import distributed
        assert self.model.training, 'call model.train() first'
        optimizer.zero_grad()
        _, _, loss = self.training_step(data, i, t_int=t_int, noise=noise, z_t=z_t, pos_encodings=pos_encodings, want_terms=False)
        loss.backward()
        optimizer.gather()
        world = distributed.allreduce_flat_gradient(optimizer.flat_grad, group=group)
        optimizer.step(grad_scale=1.0 / world)
        if update_ema and getattr(self, 'ema', None) is not None:     # (a captured step leaves it to its caller: the decay is host state)
            self.ema.update()                                                                                 # training/helpers.py:143-144
        return loss.detach()


def train_one_epoch(model, batches, optimizer, scheduler=None, *, epoch: int = 0, losses: Optional[list] = None, group=None, shuffle: bool = True):
    """training/helpers.py:121-152: one pass over `batches` (dense PlaceHolders on the device, or sparse batches graph.to_dense accepts;
    the reference's size-binned batches have different (B, n) and are taken as they come), `scheduler.step()` once per epoch, the
    epoch's average loss appended to `losses` and returned.  The running sum stays on the device: one host sync per epoch, where the
    reference syncs on every batch (`loss.cpu()`, :137)."""
    import random
    model.train()
    if shuffle:
        random.shuffle(batches)                                                                              # :127
    total = None
    for i, batch in enumerate(batches):
        dense = batch if isinstance(batch, PlaceHolder) else graph.to_dense(batch)
        loss = model.train_one_step(dense, optimizer, i, group=group)                                        # :132-144
        total = loss.clone() if total is None else total + loss
    if scheduler is not None:
        scheduler.step()                                                                                     # :146
    avg = float(total) / max(len(batches), 1) if total is not None else 0.0
    if losses is not None:
        losses.append(avg)
    return avg
