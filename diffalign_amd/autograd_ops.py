"""torch.autograd.Functions of the training path ("PyTorch-ROCm owns tensors, autograd graph, optimizer and the RCCL
process group; a thin C-ABI library owns the kernels").  Forward and backward of each Function are HIP kernels of
csrc/k_backward.hip / k_train.hip / k_elbo.hip; the dense contractions of the training forward stay nn.Linear (library GEMMs,
differentiated by autograd itself).  fp32 throughout."""
from __future__ import annotations

import math

import torch
from torch.autograd import Function

from . import _lib as L


def _c(t):
    return t.contiguous().float()


def _c16(t):
    """contiguous fp32 at a 16-byte aligned address (kernels that read float4)."""
    t = _c(t)
    return t.clone() if t.data_ptr() % 16 else t


_WS = {}


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch per device for the two-stage kernels (partial results); calls on one stream are ordered, so it is shared."""
    w = _WS.get(device)
    if w is None or w.numel() < nbytes:
        w = _WS[device] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return w


WIDE_MIN_ROWS = 4096          # below this the library's weight-gradient GEMM is as fast
_WGRAD_SHAPES = {(64, 256), (256, 64), (64, 128), (128, 64), (64, 64), (128, 128)}


class WideLinear(Function):
    """nn.Linear over the (B,n,n) pairs.  Forward and dX are library GEMMs; dW / db -- a (out,in) result reduced over B n n rows, which
    the library runs on a few workgroups -- go to da_linear_wgrad (split over rows, fp32 MFMA)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        out_f, in_f = weight.shape
        dy = _c(dy)
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        R = x.numel() // in_f
        x2, dy2 = _c(x).reshape(R, in_f), dy.reshape(R, out_f)
        dW = torch.empty_like(weight); db = torch.empty(out_f, device=weight.device)
        nbytes = L.lib().da_linear_wgrad_workspace(R, in_f, out_f)
        ws = _workspace(x.device, nbytes)
        L.check(L.lib().da_linear_wgrad(L.ptr(x2), L.ptr(dy2), L.ptr(dW), L.ptr(db), L.ptr(ws), ws.numel(), R, in_f, out_f, L.stream()),
                'da_linear_wgrad')
        return dx, dW, db


def wide_linear(lin: torch.nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """`lin(x)` for per-pair inputs: WideLinear where da_linear_wgrad is built for the shape and the row count makes it pay."""
    shape = (lin.in_features, lin.out_features)
    if shape in _WGRAD_SHAPES and x.numel() // lin.in_features >= WIDE_MIN_ROWS and lin.bias is not None and torch.is_grad_enabled():
        return WideLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


class AttnCore(Function):
    """layers.py:225-265 between the q/k/v/e_mul/e_add projections and e_out / the y_x FiLM: (Q, K, V, E1, E2, ye1, ye2) ->
    (Z = ye1 + (ye2+1) Y, wV)."""

    @staticmethod
    def forward(ctx, Q, K, V, E1, E2, nm_u8, ye1, ye2, qscale):
        Q, K, V, E1, E2, ye1, ye2 = map(_c, (Q, K, V, E1, E2, ye1, ye2))
        B, n, dx = Q.shape
        Z = torch.empty_like(E1); wV = torch.empty_like(Q); sm = torch.empty_like(Q); sl = torch.empty_like(Q)
        L.check(L.lib().da_attn_core_fwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm_u8), L.ptr(ye1), L.ptr(ye2), float(qscale),
                                         L.ptr(Z), L.ptr(wV), L.ptr(sm), L.ptr(sl), B, n, dx, L.stream()), 'da_attn_core_fwd')
        ctx.save_for_backward(Q, K, V, E1, E2, nm_u8, ye2, wV, sm, sl)
        ctx.qscale = float(qscale)
        return Z, wV

    @staticmethod
    def backward(ctx, dZ, dwV):
        Q, K, V, E1, E2, nm_u8, ye2, wV, sm, sl = ctx.saved_tensors
        B, n, dx = Q.shape
        dZ, dwV = _c(dZ), _c(dwV)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
        dE1, dE2 = torch.empty_like(E1), torch.empty_like(E1)
        dye1, dye2 = torch.empty_like(ye2), torch.empty_like(ye2)
        ws = _workspace(Q.device, L.lib().da_attn_core_bwd_workspace(B, n, dx))
        L.check(L.lib().da_attn_core_bwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm_u8), L.ptr(ye2), ctx.qscale, L.ptr(wV),
                                         L.ptr(sm), L.ptr(sl), L.ptr(dZ), L.ptr(dwV), L.ptr(dQ), L.ptr(dK), L.ptr(dV), L.ptr(dE1), L.ptr(dE2),
                                         L.ptr(dye1), L.ptr(dye2), L.ptr(ws), ws.numel(), B, n, dx, L.stream()), 'da_attn_core_bwd')
        return dQ, dK, dV, dE1, dE2, None, dye1, dye2, None


class AddLayerNorm(Function):
    """LayerNorm(a + b) (post-LN residual blocks, layers.py:341-359), eps 1e-5."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta):
        s = _c(a) + _c(b)                  # a fresh allocation: 16-byte aligned
        D = s.shape[-1]
        out = torch.empty_like(s)
        g, be = _c(gamma), _c(beta)
        L.check(L.lib().da_add_layernorm(L.ptr(s), L.ptr(None), L.ptr(g), L.ptr(be), L.ptr(out), s.numel() // D, D, L.stream()),
                'da_add_layernorm')
        ctx.save_for_backward(s, g)
        return out

    @staticmethod
    def backward(ctx, dy):
        s, g = ctx.saved_tensors
        D = s.shape[-1]
        dy = _c16(dy)
        ds = torch.empty_like(s); dg = torch.empty(D, device=s.device); db = torch.empty(D, device=s.device)
        ws = _workspace(s.device, 8 * 1024 * D)
        L.check(L.lib().da_layernorm_bwd(L.ptr(s), L.ptr(g), L.ptr(dy), L.ptr(ds), L.ptr(dg), L.ptr(db), L.ptr(ws), ws.numel(), s.numel() // D, D,
                                         L.stream()), 'da_layernorm_bwd')
        return ds, ds, dg, db


class PoolStats(Function):
    """Xtoy / Etoy statistics (layers.py:15-47): T (B, ..., d) -> hstack(mean, min, max, std) (B, 4d) over all middle axes."""

    @staticmethod
    def forward(ctx, T):
        T = _c(T)
        B, d = T.shape[0], T.shape[-1]
        N = T.numel() // (B * d)
        z = torch.empty(B, 4, d, device=T.device); arg = torch.empty(B, 2, d, dtype=torch.int32, device=T.device)
        ws = _workspace(T.device, 32 * B * 64 * d) if N > 256 else None
        L.check(L.lib().da_pool_stats_fwd(L.ptr(T), L.ptr(z), L.ptr(arg), L.ptr(ws), ws.numel() if ws is not None else 0, B, N, d, L.stream()),
                'da_pool_stats_fwd')
        ctx.save_for_backward(T, z, arg)
        return z.reshape(B, 4 * d)

    @staticmethod
    def backward(ctx, dz):
        T, z, arg = ctx.saved_tensors
        B, d = T.shape[0], T.shape[-1]
        N = T.numel() // (B * d)
        dT = torch.empty_like(T)
        L.check(L.lib().da_pool_stats_bwd(L.ptr(T), L.ptr(z), L.ptr(arg), L.ptr(_c(dz)), L.ptr(dT), B, N, d, L.stream()), 'da_pool_stats_bwd')
        return dT


class MaskDense(Function):
    """PlaceHolder.mask() on hidden features (transformer_model_with_y.py:661): X *= m_i, E *= m_i m_j, E[diag] = 0, then
    encode_no_element.  The gradient passes where the value passed (the constant 1 of encode_no_element carries none)."""

    @staticmethod
    def forward(ctx, X, E, nm_u8):
        X, E = _c(X).clone(), _c(E).clone()
        B, n = nm_u8.shape
        L.check(L.lib().da_mask_dense(L.ptr(X), L.ptr(E), L.ptr(nm_u8), B, n, X.shape[-1], E.shape[-1], L.stream()), 'da_mask_dense')
        ctx.save_for_backward(nm_u8)
        return X, E

    @staticmethod
    def backward(ctx, dX, dE):
        (nm_u8,) = ctx.saved_tensors
        m = nm_u8.bool()
        n = m.shape[1]
        pm = m[:, :, None] & m[:, None, :] & ~torch.eye(n, dtype=torch.bool, device=m.device)[None]
        return dX * m[:, :, None], dE * pm[..., None], None


class DiffusionLoss(Function):
    """Training objective on the denoiser's logits (diffusion_abstract.py:198-222): 'ce' (helpers.py:153-191), 'vb'
    (elbo_batch_quick, elbo.py:273-334) or 'vbce' = vb + ce_lambda * ce.  The value comes from the ELBO row kernels, the gradient
    from da_ce_grad / da_vb_grad; `host` is the diffusion object (transition tables, cfg), everything else plain tensors."""

    @staticmethod
    def forward(ctx, lX, lE, host, dense_true, z_t, node_masks, kind):
        from . import elbo as E_
        from .placeholder import PlaceHolder
        d = host.cfg.diffusion
        lam = float(d.lambda_train[0])
        pred = PlaceHolder(X=_c(lX.detach()), E=_c(lE.detach()), y=None, node_mask=dense_true.node_mask)
        loss = torch.zeros((), device=lX.device)
        if kind in ('ce', 'vbce'):
            rX, rE = E_.elbo_rows(E_.LOGP, dense_true, logits=pred)
            s = E_.masked_sums(rX, rE, node_masks).double()
            ce = (-(s[:, 0].sum() / s[:, 1].sum()) * float(bool(d.diffuse_nodes))
                  - float(bool(d.diffuse_edges)) * lam * (s[:, 2].sum() / s[:, 3].sum())).float()
            loss = loss + (ce if kind == 'ce' else float(d.ce_lambda) * ce)
            ctx.ce_rows = (rX, rE)
        if kind in ('vb', 'vbce'):
            loss = loss + host.elbo_batch_quick(dense_true, pred=pred, z_t=z_t, lambda_E=lam)
        ctx.host, ctx.dense_true, ctx.z_t, ctx.node_masks, ctx.kind, ctx.pred = host, dense_true, z_t, node_masks, kind, pred
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import elbo as E_
        host, true, z_t, masks, kind, pred = ctx.host, ctx.dense_true, ctx.z_t, ctx.node_masks, ctx.kind, ctx.pred
        d = host.cfg.diffusion
        lam = float(d.lambda_train[0])
        B, n, kx = pred.X.shape
        ke = pred.E.shape[-1]
        dev = pred.X.device
        mx, me = L.as_u8(masks[0]), L.as_u8(masks[1])
        sums = E_.masked_sums(torch.zeros(B, n, device=dev), torch.zeros(B, n, n, device=dev), masks)
        tX, tE = _c(true.X), _c(true.E)
        gX = torch.zeros_like(pred.X); gE = torch.zeros_like(pred.E)
        wn, we = float(bool(d.diffuse_nodes)), float(bool(d.diffuse_edges)) * lam
        if kind in ('ce', 'vbce'):
            w = 1.0 if kind == 'ce' else float(d.ce_lambda)
            cX, cE = torch.empty_like(gX), torch.empty_like(gE)
            L.check(L.lib().da_ce_grad(L.ptr(tX), L.ptr(tE), L.ptr(pred.X), L.ptr(pred.E), L.ptr(mx), L.ptr(me), L.ptr(sums), w * wn, w * we,
                                       L.ptr(cX), L.ptr(cE), B, n, kx, ke, L.stream()), 'da_ce_grad')
            gX += cX; gE += cE
        if kind in ('vb', 'vbce'):
            t = z_t.y
            tm = host.transition_model
            Qt, Qsb = tm.get_Qt(t), tm.get_Qt_bar(t - 1)
            t1 = (t[:, 0] == 1).to(torch.uint8).contiguous()
            T_ = float(d.diffusion_steps)
            vX, vE = torch.empty_like(gX), torch.empty_like(gE)
            L.check(L.lib().da_vb_grad(L.ptr(tX), L.ptr(tE), L.ptr(_c(z_t.X)), L.ptr(_c(z_t.E)), L.ptr(pred.X), L.ptr(pred.E), L.ptr(_c(Qt.X)),
                                       L.ptr(_c(Qsb.X)), L.ptr(_c(Qt.E)), L.ptr(_c(Qsb.E)), L.ptr(t1), L.ptr(mx), L.ptr(me), L.ptr(sums),
                                       T_ * wn, T_ * we, L.ptr(vX), L.ptr(vE), B, n, kx, ke, L.stream()), 'da_vb_grad')
            gX += vX; gE += vE
        return gX * g, gE * g, None, None, None, None, None


def qscale_of(dx: int, n_head: int) -> float:
    return 1.0 / math.sqrt(dx // n_head)
