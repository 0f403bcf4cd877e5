"""ELBOMixin with the reference's method names (diffalign/diffusion/elbo.py) on the HIP library.

Every per-node / per-pair KL or log-likelihood map is ONE bandwidth-bound kernel launch (da_elbo_rows) and the masked
means are a per-graph reduction on the device (da_masked_sums): the reference's `.to('cpu')` of every map
(elbo.py:65,69) is gone.  Free masks are per-node vectors (graph.get_node_masks), never the dense (B,n,n,K) tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L, graph
from .placeholder import PlaceHolder

LT_PROB, LT_LOG, KL_PRIOR, RECON, LOGP = range(5)


def _node_masks(model, dense_true):
    d = model.cfg.diffusion
    return graph.get_node_masks(dense_true, model.dataset_info.atom_decoder, model.dataset_info.bond_decoder, d.mask_nodes,
                                d.mask_edges, node_states_to_mask=list(getattr(d, 'node_states_to_mask', []) or []))


def elbo_rows(mode: int, true: PlaceHolder, z_t: Optional[PlaceHolder] = None, logits: Optional[PlaceHolder] = None,
              Qa: Optional[PlaceHolder] = None, Qb: Optional[PlaceHolder] = None, limit: Optional[PlaceHolder] = None,
              node_masks=None, eps: float = 1e-6):
    """Rows (B,n) and (B,n,n) of one ELBO term; see csrc/k_elbo.hip for the modes."""
    X, E = true.X.contiguous().float(), true.E.contiguous().float()
    B, n, kx = X.shape
    ke = E.shape[-1]
    dev = X.device
    f = lambda t: t.contiguous().float() if t is not None else None
    zx, ze = (f(z_t.X), f(z_t.E)) if z_t is not None else (None, None)
    lx, le = (f(logits.X), f(logits.E)) if logits is not None else (None, None)
    qax, qae = (f(Qa.X), f(Qa.E)) if Qa is not None else (None, None)
    qbx, qbe = (f(Qb.X), f(Qb.E)) if Qb is not None else (None, None)
    per_graph = 1 if (qax is not None and qax.dim() == 3) else 0
    limx, lime = (f(limit.X).to(dev), f(limit.E).to(dev)) if limit is not None else (None, None)
    mx, me = (L.as_u8(node_masks[0]), L.as_u8(node_masks[1])) if node_masks is not None else (None, None)
    rX = torch.empty(B, n, device=dev)
    rE = torch.empty(B, n, n, device=dev)
    L.check(L.lib().da_elbo_rows(C.c_int(mode), L.ptr(X), L.ptr(E), L.ptr(zx), L.ptr(ze), L.ptr(lx), L.ptr(le), L.ptr(qax),
                                 L.ptr(qbx), L.ptr(qae), L.ptr(qbe), C.c_int(per_graph), L.ptr(limx), L.ptr(lime), L.ptr(mx),
                                 L.ptr(me), C.c_float(eps), L.ptr(rX), L.ptr(rE), C.c_int(B), C.c_int(n), C.c_int(kx),
                                 C.c_int(ke), L.stream()), 'da_elbo_rows')
    return rX, rE


def masked_sums(tX, tE, node_masks):
    """(B,4): sum over free nodes, #free nodes, sum over free pairs, #free pairs."""
    B, n = tX.shape
    out = torch.empty(B, 4, device=tX.device)
    L.check(L.lib().da_masked_sums(L.ptr(tX.contiguous().float()), L.ptr(tE.contiguous().float()), L.ptr(L.as_u8(node_masks[0])),
                                   L.ptr(L.as_u8(node_masks[1])), L.ptr(out), C.c_int(B), C.c_int(n), L.stream()),
            'da_masked_sums')
    return out


def mean_without_masked(graph_obj: PlaceHolder, node_masks, diffuse_edges=True, diffuse_nodes=True, avg_over_batch=False):
    """helpers.mean_without_masked (utils/diffusion/helpers.py:59-106) with per-node free masks.  graph_obj.X (B,n),
    graph_obj.E (B,n,n)."""
    s = masked_sums(graph_obj.X, graph_obj.E, node_masks).double()
    if avg_over_batch:
        mX, mE = s[:, 0].sum() / s[:, 1].sum(), s[:, 2].sum() / s[:, 3].sum()
    else:
        mX, mE = s[:, 0] / s[:, 1], s[:, 2] / s[:, 3]
    if diffuse_nodes and diffuse_edges:
        return (mX + mE).float()
    return (mE if diffuse_edges else mX).float()


def reconstruction_logp(orig: PlaceHolder, pred_logits: PlaceHolder, node_masks):
    """helpers.reconstruction_logp (helpers.py:139-151) fused with the apply_mask(as_logits=True) + log_softmax that
    precede it in compute_L1 (elbo.py:182-191)."""
    rX, rE = elbo_rows(RECON, orig, logits=pred_logits, node_masks=node_masks)
    return orig.get_new_object(X=rX, E=rE)


def ce(pred: PlaceHolder, dense_true: PlaceHolder, node_masks, lambda_E: float = 5.0):
    """helpers.ce (helpers.py:153-191): mean cross-entropy over free nodes + lambda_E * mean over free pairs."""
    rX, rE = elbo_rows(LOGP, dense_true, logits=pred)
    s = masked_sums(rX, rE, node_masks).double()
    return (-(s[:, 0].sum() / s[:, 1].sum()) - lambda_E * (s[:, 2].sum() / s[:, 3].sum())).float()


class ELBOMixin:
    """Needs from the host object: cfg, T, eps, transition_model(_eval), limit_dist, dataset_info, forward, apply_noise,
    get_pos_encodings_if_relevant (diffusion_abstract.py)."""

    def compute_Lt(self, dense_true, z_t, t, x_0_tilde_logit, transition_model, log=False, node_masks=None):
        """elbo.py:85-162.  Returns PlaceHolder with X (B,n), E (B,n,n) = KL(q(z_s|z_t,x_0) || p(z_s|z_t))."""
        assert t.shape[1] == 1, 't should be a tensor of shape (bs, 1)'
        if node_masks is None:
            node_masks = _node_masks(self, dense_true)
        Qt = transition_model.get_Qt(t)
        Qsb = transition_model.get_Qt_bar(t - 1)
        kX, kE = elbo_rows(LT_LOG if log else LT_PROB, dense_true, z_t=z_t, logits=x_0_tilde_logit, Qa=Qt, Qb=Qsb,
                           node_masks=node_masks, eps=self.eps)
        return z_t.get_new_object(X=kX, E=kE)

    def _noised(self, dense_true, t_int, node_masks, noise=None):
        z_t, _ = self.apply_noise(dense_true, t_int=t_int, transition_model=self.transition_model_eval, noise=noise)
        z_t.y = z_t.y * (self.T // self.cfg.diffusion.diffusion_steps_eval)
        return graph.apply_mask(orig=dense_true, z_t=z_t, atom_decoder=self.dataset_info.atom_decoder,
                                bond_decoder=self.dataset_info.bond_decoder, mask_nodes=self.cfg.diffusion.mask_nodes,
                                mask_edges=self.cfg.diffusion.mask_edges, node_masks=node_masks)

    def compute_Lt_all(self, dense_true, reduce_with=None, noise_fn=None, node_masks=None):
        """elbo.py:16-83.  reduce_with=None: list of per-step KL PlaceHolders (kept on the device); otherwise the running
        sum of the masked means (a (B,) tensor or a scalar).  noise_fn() -> (noise_X, noise_E) Exp(1) tensors or None."""
        d = self.cfg.diffusion
        assert self.T % d.diffusion_steps_eval == 0, 'diffusion_steps_eval should be divisible by diffusion_steps'
        if node_masks is None:
            node_masks = _node_masks(self, dense_true)
        B = dense_true.X.shape[0]
        dev = dense_true.X.device
        pos_encodings = self.get_pos_encodings_if_relevant(dense_true)
        Lts, running = ([] if reduce_with is None else None), None
        for t in range(2, d.diffusion_steps_eval + 1):
            t_int = torch.full((B, 1), float(t), device=dev)
            z_t = self._noised(dense_true, t_int, node_masks, noise_fn() if noise_fn else None)
            pred = self.forward(z_t=z_t, pos_encodings=pos_encodings)
            Lt = self.compute_Lt(dense_true=dense_true, z_t=z_t, t=t_int, x_0_tilde_logit=pred,
                                 transition_model=self.transition_model_eval, node_masks=node_masks)
            if reduce_with is None:
                Lts.append(Lt)
            else:
                Lt.E = Lt.E * reduce_with['lambda_E']
                term = mean_without_masked(Lt, node_masks, diffuse_nodes=reduce_with['diffuse_nodes'],
                                           diffuse_edges=reduce_with['diffuse_edges'], avg_over_batch=reduce_with['avg_over_batch'])
                running = term if running is None else running + term
        return Lts if reduce_with is None else running

    def compute_L1(self, dense_true, pos_encodings=None, noise=None, node_masks=None):
        """elbo.py:164-195: reconstruction term log p(x_0 | z_1)."""
        if node_masks is None:
            node_masks = _node_masks(self, dense_true)
        B = dense_true.X.shape[0]
        t_int = torch.ones((B, 1), device=dense_true.X.device)
        z_1 = self._noised(dense_true, t_int, node_masks, noise)
        pred0 = self.forward(z_t=z_1, pos_encodings=pos_encodings)
        return reconstruction_logp(dense_true, pred0, node_masks)

    def kl_prior(self, dense_true, node_masks=None):
        """elbo.py:197-238 + helpers.kl_prior (helpers.py:115-137)."""
        if node_masks is None:
            node_masks = _node_masks(self, dense_true)
        B = dense_true.X.shape[0]
        Ts = self.T * torch.ones((1, 1), device=dense_true.X.device)
        Qtb = self.transition_model.get_Qt_bar(Ts)
        Qa = PlaceHolder(X=Qtb.X[0], E=Qtb.E[0], y=None)
        kX, kE = elbo_rows(KL_PRIOR, dense_true, Qa=Qa, limit=self.limit_dist, node_masks=node_masks, eps=self.eps)
        return dense_true.get_new_object(X=kX, E=kE)

    @torch.no_grad()
    def elbo(self, dense_true, avg_over_batch=True, noise_fn=None):
        """elbo.py:336-405: vb = KL prior + sum_t L_t - L_0, each averaged over free nodes + lambda_test * free pairs."""
        d = self.cfg.diffusion
        node_masks = _node_masks(self, dense_true)
        mw = lambda obj: mean_without_masked(obj, node_masks, diffuse_edges=d.diffuse_edges, diffuse_nodes=d.diffuse_nodes,
                                             avg_over_batch=avg_over_batch)
        kl_prior = self.kl_prior(dense_true, node_masks)
        loss_t = self.compute_Lt_all(dense_true, reduce_with={'lambda_E': d.lambda_test, 'avg_over_batch': avg_over_batch,
                                                              'diffuse_nodes': d.diffuse_nodes, 'diffuse_edges': d.diffuse_edges},
                                     noise_fn=noise_fn, node_masks=node_masks)
        pos_encodings = self.get_pos_encodings_if_relevant(dense_true)
        loss_0s = []
        for _ in range(self.cfg.test.loss_0_repeat):
            l0 = self.compute_L1(dense_true=dense_true, pos_encodings=pos_encodings, noise=noise_fn() if noise_fn else None,
                                 node_masks=node_masks)
            l0.E = l0.E * d.lambda_test
            loss_0s.append(mw(l0))
        loss_0 = sum(loss_0s) / self.cfg.test.loss_0_repeat
        kl_prior_per_dim = mw(kl_prior)
        if loss_t is None:
            loss_t = torch.zeros_like(kl_prior_per_dim)
        return kl_prior_per_dim + loss_t - loss_0, loss_t, loss_0

    def elbo_batch_quick(self, dense_true, pred, z_t, lambda_E=1.0, avg_over_batch=True):
        """elbo.py:273-334: one-timestep-per-graph estimator of the bound (training objective 'vb'), forward value."""
        d = self.cfg.diffusion
        node_masks = _node_masks(self, dense_true)
        t = z_t.y
        term_t = self.compute_Lt(dense_true=dense_true, z_t=z_t, t=t, x_0_tilde_logit=pred,
                                 transition_model=self.transition_model, log=True, node_masks=node_masks)
        kl_prior = self.kl_prior(dense_true, node_masks)
        rX, rE = elbo_rows(LOGP, dense_true, logits=pred)              # reconstruction_logp of log_softmax(pred)
        is1 = (t[:, 0] == 1)
        tX = torch.where(is1[:, None], -rX, term_t.X)
        tE = torch.where(is1[:, None, None], -rE, term_t.E) * lambda_E
        mw = lambda X_, E_: mean_without_masked(PlaceHolder(X=X_, E=E_, y=None), node_masks, diffuse_edges=d.diffuse_edges,
                                                diffuse_nodes=d.diffuse_nodes, avg_over_batch=avg_over_batch)
        return mw(tX, tE) * d.diffusion_steps + mw(kl_prior.X, kl_prior.E * lambda_E)
