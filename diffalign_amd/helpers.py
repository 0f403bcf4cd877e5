"""Diffusion math with the reference's function names (diffalign/utils/diffusion/helpers.py), executed by the
HIP library.  Categorical draws take explicit Exp(1) "race" noise (argmax_k p_k/q_k == torch.multinomial's CPU
algorithm); when no noise tensor is given it is drawn on the device with torch's generator."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L
from .placeholder import PlaceHolder


def exp_noise_like(shape, device, generator=None) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=device).exponential_(1.0, generator=generator)


def counter_exp_noise(chain_seeds: torch.Tensor, call: int, B, n, kx, ke):
    """Exp(1) race noise (B,n,kx), (B,n,n,ke) as a pure function of (chain seed, draw call, element): a chain's draws do not
    depend on which batch or rank it runs in (C-ABI da_counter_exp_noise; distributed.chain_seed gives the seeds)."""
    dev = chain_seeds.device
    qX = torch.empty(B, n, kx, device=dev); qE = torch.empty(B, n, n, ke, device=dev)
    L.check(L.lib().da_counter_exp_noise(L.ptr(chain_seeds.contiguous(), torch.int64), int(call), L.ptr(qX), L.ptr(qE), B, n, kx, ke,
                                         L.stream()), 'da_counter_exp_noise')
    return qX, qE


def posterior_sample(pred_X, pred_E, z_t: PlaceHolder, Qt: PlaceHolder, Qsb: PlaceHolder, Qtb: PlaceHolder,
                     temperature=(1.0, 1.0), noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                     orig: Optional[PlaceHolder] = None, node_masks=None, want_prob=False, want_sample=True,
                     generator=None):
    """Fused reverse step on raw logits (C-ABI da_posterior_sample).  Returns (prob_X, prob_E, zs_X, zs_E)."""
    B, n, kx = z_t.X.shape
    ke = z_t.E.shape[-1]
    dev = z_t.X.device
    f = lambda t: t.contiguous().float()
    pX, pE, ztX, ztE = f(pred_X), f(pred_E), f(z_t.X), f(z_t.E)
    probX = torch.empty(B, n, kx, device=dev) if want_prob else None
    probE = torch.empty(B, n, n, ke, device=dev) if want_prob else None
    zsX = torch.empty(B, n, kx, device=dev) if want_sample else None
    zsE = torch.empty(B, n, n, ke, device=dev) if want_sample else None
    qX = qE = None
    if want_sample:
        if noise is None:
            noise = (exp_noise_like((B, n, kx), dev, generator), exp_noise_like((B, n, n, ke), dev, generator))
        qX, qE = f(noise[0]), f(noise[1])
    oX = oE = mx = me = None
    if node_masks is not None:
        assert orig is not None
        oX, oE = f(orig.X), f(orig.E)
        mx, me = L.as_u8(node_masks[0]), L.as_u8(node_masks[1])
    tabs = [f(Qt.X), f(Qsb.X), f(Qtb.X), f(Qt.E), f(Qsb.E), f(Qtb.E)]
    nm = L.as_u8(z_t.node_mask)

    def launch(node_part, edge_part):
        n_ = (lambda t: t) if node_part else (lambda t: None)
        e_ = (lambda t: t) if edge_part else (lambda t: None)
        L.check(L.lib().da_posterior_sample(
            L.ptr(pX), L.ptr(pE), L.ptr(ztX), L.ptr(ztE), L.ptr(nm),
            *[L.ptr(t) for t in tabs], C.c_float(temperature[0]), C.c_float(temperature[1]),
            L.ptr(n_(qX)), L.ptr(e_(qE)), L.ptr(oX), L.ptr(oE), L.ptr(n_(mx)), L.ptr(e_(me)),
            L.ptr(n_(probX)), L.ptr(e_(probE)), L.ptr(n_(zsX)), L.ptr(e_(zsE)),
            C.c_int(B), C.c_int(n), C.c_int(kx), C.c_int(ke), L.stream()), 'da_posterior_sample')

    from . import engine
    sides = engine._side_stream(dev)
    if sides is None:
        launch(True, True)
    else:
        # the node half (few workgroups, latency-bound) beside the edge half; every tensor was allocated on the main stream
        # and the main stream joins before anything is returned
        main = torch.cuda.current_stream()
        ready = main.record_event()
        with torch.cuda.stream(sides[1]):
            sides[1].wait_event(ready)
            launch(True, False)
            done = sides[1].record_event()
        launch(False, True)
        main.wait_event(done)
    return probX, probE, zsX, zsE


def reverse_step_ctx(z_t: PlaceHolder, Qt: PlaceHolder, Qsb: PlaceHolder, Qtb: PlaceHolder, temperature=(1.0, 1.0),
                     noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, chain_seeds: Optional[torch.Tensor] = None, call: int = 0,
                     orig: Optional[PlaceHolder] = None, node_masks=None, generator=None, out=None, out_holds_condition=False, last_rows=None) -> dict:
    """What engine.model_forward(step=...) needs to draw z_s behind the denoiser's output stage (the fused form of posterior_sample).
    Noise, in this order of preference: explicit Exp(1) tensors `noise` = (qX, qE); per-chain counter noise (`chain_seeds`, `call`:
    the edge half generates da_counter_exp_noise's values in registers, never materialising the (B,n,n,ke) tensor); otherwise
    per-chain seeds are drawn from torch's generator and used the same way."""
    B, n, kx = z_t.X.shape
    ke = z_t.E.shape[-1]
    dev = z_t.X.device
    f = lambda t: t.contiguous().float()
    qX = qE = seeds = None
    if noise is not None:
        qX, qE = f(noise[0]), f(noise[1])
    else:
        if chain_seeds is None:
            chain_seeds = torch.randint(0, 2 ** 62, (B,), device=dev, generator=generator)
            call = 0
        seeds = chain_seeds.to(device=dev, dtype=torch.int64).contiguous()
        qX = torch.empty(B, n, kx, device=dev)
        # the node half's noise is read at the END of the step (da_posterior_sample behind the node output path): generated on the y
        # stream, beside the input stage, instead of in front of it (the caller's stream joins that stream before the first layer)
        from . import engine
        sides = engine._side_stream(dev)
        if sides is not None and not torch.cuda.is_current_stream_capturing():   # (captured steps keep it on the caller's stream: a
            # fork this early in the captured step made the graph's replay fault)
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(sides[0]):
                sides[0].wait_event(ev)
                L.check(L.lib().da_counter_exp_noise(L.ptr(seeds, torch.int64), int(call), L.ptr(qX), L.ptr(None), B, n, kx, ke, L.stream()),
                        'da_counter_exp_noise')
        else:
            L.check(L.lib().da_counter_exp_noise(L.ptr(seeds, torch.int64), int(call), L.ptr(qX), L.ptr(None), B, n, kx, ke, L.stream()),
                    'da_counter_exp_noise')
    if out is not None:          # the caller's buffers receive z_s (graphs.GraphedChain: the chain's state alternates between two holders)
        zsX, zsE = out
        assert zsX.shape == (B, n, kx) and zsE.shape == (B, n, n, ke) and zsX.dtype == zsE.dtype == torch.float32
        assert zsX.is_contiguous() and zsE.is_contiguous() and zsE.data_ptr() != z_t.E.data_ptr() and zsX.data_ptr() != z_t.X.data_ptr()
    else:
        zsX, zsE = torch.empty(B, n, kx, device=dev), torch.empty(B, n, n, ke, device=dev)
    oX = oE = mx = me = None
    if node_masks is not None:
        assert orig is not None
        oX, oE = f(orig.X), f(orig.E)
        mx, me = L.as_u8(node_masks[0]), L.as_u8(node_masks[1])
    return dict(zt_X=f(z_t.X), zt_E=f(z_t.E), Qt_x=f(Qt.X), Qsb_x=f(Qsb.X), Qtb_x=f(Qtb.X), Qt_e=f(Qt.E), Qsb_e=f(Qsb.E), Qtb_e=f(Qtb.E),
                temp=(float(temperature[0]), float(temperature[1])), noise_X=qX, noise_E=qE, seeds=seeds, call=int(call),
                orig_X=oX, orig_E=oE, mx=mx, me=me, zs_X=zsX, zs_E=zsE, keep_fixed=bool(out_holds_condition and out is not None),
                last_rows=last_rows if node_masks is not None else None)


def get_p_zs_given_zt(transition_model, t_array, pred, z_t, return_prob=False, temperature_scaling=1.0, noise=None):
    """helpers.py:193-245.  `pred` holds logits (the caller may have applied log_softmax*temperature already, as
    diffusion/sampling.py:82 does: softmax is idempotent on log-probabilities)."""
    dev = z_t.X.device
    Qtb = transition_model.get_Qt_bar(t_array, dev)
    Qsb = transition_model.get_Qt_bar(t_array - 1, dev)
    Qt = transition_model.get_Qt(t_array, dev)
    probX, probE, zsX, zsE = posterior_sample(pred.X, pred.E, z_t, Qt, Qsb, Qtb, noise=noise, want_prob=return_prob,
                                              want_sample=not return_prob)
    if return_prob:
        return z_t.get_new_object(X=probX, E=probE)
    out = z_t.get_new_object(X=zsX, E=zsE, y=torch.zeros(z_t.y.shape[0], 0, device=dev))
    disc = out.get_new_object().mask(z_t.node_mask, collapse=True)
    return out.mask(z_t.node_mask).type_as(z_t.y), disc.type_as(z_t.y)


def sample_discrete_features(prob: PlaceHolder, noise=None, generator=None, return_indices=False):
    """helpers.py:706-752 (C-ABI da_sample_discrete); bit-exact for identical prob and noise."""
    B, n, kx = prob.X.shape
    ke = prob.E.shape[-1]
    dev = prob.X.device
    if noise is None:
        noise = (exp_noise_like((B, n, kx), dev, generator), exp_noise_like((B, n, n, ke), dev, generator))
    pX, pE = prob.X.contiguous().float(), prob.E.contiguous().float()
    Xs = torch.empty(B, n, dtype=torch.int64, device=dev)
    Es = torch.empty(B, n, n, dtype=torch.int64, device=dev)
    X1 = torch.empty(B, n, kx, device=dev); E1 = torch.empty(B, n, n, ke, device=dev)
    L.check(L.lib().da_sample_discrete(L.ptr(pX), L.ptr(pE), L.ptr(L.as_u8(prob.node_mask)),
                                       L.ptr(noise[0].contiguous().float()), L.ptr(noise[1].contiguous().float()),
                                       L.ptr(Xs), L.ptr(Es), L.ptr(X1), L.ptr(E1), C.c_int(B), C.c_int(n), C.c_int(kx),
                                       C.c_int(ke), L.stream()), 'da_sample_discrete')
    z = prob.get_new_object(X=X1, E=E1, y=prob.y)
    return (z, Xs, Es) if return_indices else z


def noise_probs(dense_data: PlaceHolder, Qtb: PlaceHolder):
    """diffusion_abstract.py:253-257: prob = data @ Qbar_t, .mask()."""
    B, n, kx = dense_data.X.shape
    ke = dense_data.E.shape[-1]
    dev = dense_data.X.device
    pX = torch.empty(B, n, kx, device=dev); pE = torch.empty(B, n, n, ke, device=dev)
    L.check(L.lib().da_noise_probs(L.ptr(dense_data.X.contiguous().float()), L.ptr(dense_data.E.contiguous().float()),
                                   L.ptr(L.as_u8(dense_data.node_mask)), L.ptr(Qtb.X.contiguous().float()),
                                   L.ptr(Qtb.E.contiguous().float()), L.ptr(pX), L.ptr(pE), C.c_int(B), C.c_int(n),
                                   C.c_int(kx), C.c_int(ke), L.stream()), 'da_noise_probs')
    return pX, pE


def sample_from_noise(limit_dist, node_mask, T, noise=None, generator=None):
    """helpers.py:831-859: z_T from the limit distribution (draw per node / per pair, upper triangle mirrored,
    .mask()).  Same kernel as sample_discrete_features: the limit distribution is broadcast to every element."""
    B, n = node_mask.shape
    dev = node_mask.device
    kx, ke = limit_dist.X.numel(), limit_dist.E.numel()
    probX = limit_dist.X.to(dev).float()[None, None, :].expand(B, n, -1).contiguous()
    probE = limit_dist.E.to(dev).float()[None, None, None, :].expand(B, n, n, -1).contiguous()
    full = torch.ones_like(node_mask)
    ph = PlaceHolder(X=probX, E=probE, y=T * torch.ones((B, 1), device=dev), node_mask=full)
    z = sample_discrete_features(ph, noise=noise, generator=generator)
    return PlaceHolder(X=z.X, E=z.E, y=z.y, node_mask=node_mask).mask(node_mask)
