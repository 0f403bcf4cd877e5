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
    """Grow-only scratch per (device, stream) for the two-stage kernels (partial results): calls on one stream are ordered, so they
    share it; the node and pair sides of the training step run on two streams (train_forward.py) and must not.  A buffer that is
    outgrown is handed back through the allocator, which defers its reuse behind the stream's pending work."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _WS.get(key)
    if w is None or w.numel() < nbytes:
        w = _WS[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return w


WIDE_MIN_ROWS = 4096          # below this the library's weight-gradient GEMM is as fast
_WGRAD_SHAPES = {(64, 256), (256, 64), (64, 128), (128, 64), (64, 64), (128, 128)}


class WideLinear(Function):
    """nn.Linear over the (B,n,n) pairs.  Forward and dX are library GEMMs; dW / db -- a (out,in) result reduced over B n n rows, which
    the library runs on a few workgroups -- go to da_linear_wgrad (split over rows, fp32 MFMA)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act=None):
        out = torch.nn.functional.linear(x, weight, bias)
        ctx.act = act is not None
        if act is not None:                     # dropout_p(relu(.)) in place on the result (act = (seed, salt, p): da_relu_dropout)
            _relu_dropout_(out, *act)
            ctx.p = float(act[2])
            ctx.save_for_backward(x, weight, out)
        else:
            ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors[:2]
        out_f, in_f = weight.shape
        dy = _c(dy)
        if ctx.act:
            dy = _relu_dropout_bwd(ctx.saved_tensors[2], dy, ctx.p)
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        R = x.numel() // in_f
        x2, dy2 = _c(x).reshape(R, in_f), dy.reshape(R, out_f)
        dW = torch.empty_like(weight); db = torch.empty(out_f, device=weight.device)
        nbytes = L.lib().da_linear_wgrad_workspace(R, in_f, out_f)
        ws = _workspace(x.device, nbytes)
        L.check(L.lib().da_linear_wgrad(L.ptr(x2), L.ptr(dy2), L.ptr(dW), L.ptr(db), L.ptr(ws), ws.numel(), R, in_f, out_f, L.stream()),
                'da_linear_wgrad')
        return dx, dW, db, None


class WideLinearBF16(Function):
    """WideLinear in mixed precision (the reference trains under fp16 autocast, diffusion_abstract.py:181-183): x, the weight copy and the
    result are bf16 (library GEMM, fp32 accumulation), dX is a bf16 GEMM, dW / db come from da_linear_wgrad_mixed -- bf16 operands
    widened on load, fp32 products and reduction -- as fp32 gradients of the fp32 master weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, act=None):
        wb = weight.to(torch.bfloat16)
        out = torch.nn.functional.linear(x, wb, bias.to(torch.bfloat16))
        ctx.act = act is not None
        if act is not None:
            _relu_dropout_(out, *act)
            ctx.p = float(act[2])
            ctx.save_for_backward(x, wb, out)
        else:
            ctx.save_for_backward(x, wb)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors[:2]
        out_f, in_f = wb.shape
        dy = dy.contiguous()
        if ctx.act:
            dy = _relu_dropout_bwd(ctx.saved_tensors[2], dy, ctx.p)
        dx = dy @ wb if ctx.needs_input_grad[0] else None
        R = x.numel() // in_f
        x2, dy2 = x.contiguous().reshape(R, in_f), dy.reshape(R, out_f)
        dW = torch.empty(out_f, in_f, device=x.device); db = torch.empty(out_f, device=x.device)
        ws = _workspace(x.device, L.lib().da_linear_wgrad_workspace(R, in_f, out_f))
        L.check(L.lib().da_linear_wgrad_mixed(L.ptr(x2), int(x2.dtype == torch.bfloat16), L.ptr(dy2), int(dy2.dtype == torch.bfloat16), L.ptr(dW),
                                              L.ptr(db), L.ptr(ws), ws.numel(), R, in_f, out_f, L.stream()), 'da_linear_wgrad_mixed')
        return dx, dW, db, None


def _relu_dropout_(h, seed, salt, p):
    p = float(p)
    L.check(L.lib().da_relu_dropout(L.ptr(h), int(h.dtype == torch.bfloat16), h.numel(), L.ptr(seed if p > 0 else None), salt, p, L.stream()),
            'da_relu_dropout')
    return h


def _relu_dropout_bwd(h, dy, p):
    dy = dy.contiguous()
    if dy.dtype != h.dtype:
        dy = dy.to(h.dtype)
    dx = torch.empty_like(h)
    L.check(L.lib().da_relu_dropout_bwd(L.ptr(h), L.ptr(dy), L.ptr(dx), int(h.dtype == torch.bfloat16), h.numel(), float(p), L.stream()),
            'da_relu_dropout_bwd')
    return dx


def wide_linear(lin: torch.nn.Linear, x: torch.Tensor, act=None) -> torch.Tensor:
    """`lin(x)` for per-pair inputs: WideLinear where da_linear_wgrad is built for the shape and the row count makes it pay.  A bf16
    input selects the mixed-precision form (bf16 in, bf16 out).  `act` = (seed, salt, p): dropout_p(relu(lin(x))), applied in place on
    the GEMM's result (csrc/k_train_fused.hip)."""
    shape = (lin.in_features, lin.out_features)
    if x.dtype == torch.bfloat16:
        if shape in _WGRAD_SHAPES and lin.bias is not None and torch.is_grad_enabled():
            return WideLinearBF16.apply(x, lin.weight, lin.bias, act)
        out = torch.nn.functional.linear(x, lin.weight.to(torch.bfloat16), None if lin.bias is None else lin.bias.to(torch.bfloat16))
    elif shape in _WGRAD_SHAPES and x.numel() // lin.in_features >= WIDE_MIN_ROWS and lin.bias is not None and torch.is_grad_enabled():
        return WideLinear.apply(x, lin.weight, lin.bias, act)
    else:
        out = lin(x)
    return out if act is None else ReluDropout.apply(out, *act)


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


class AttnCoreBF16(Function):
    """AttnCore with the per-pair tensors in bf16 storage: E1, E2 in, Z out, and dZ in, dE1, dE2 out in the backward; q, k, v, the FiLM
    vectors, the softmax statistics and all arithmetic stay fp32 (da_attn_core_fwd_bf16 / _bwd_bf16)."""

    @staticmethod
    def forward(ctx, Q, K, V, E1, E2, nm_u8, ye1, ye2, qscale):
        Q, K, V, ye1, ye2 = map(_c, (Q, K, V, ye1, ye2))
        E1, E2 = E1.contiguous(), E2.contiguous()
        assert E1.dtype == torch.bfloat16 and E2.dtype == torch.bfloat16
        B, n, dx = Q.shape
        Z = torch.empty_like(E1); wV = torch.empty_like(Q); sm = torch.empty_like(Q); sl = torch.empty_like(Q)
        L.check(L.lib().da_attn_core_fwd_bf16(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm_u8), L.ptr(ye1), L.ptr(ye2), float(qscale),
                                              L.ptr(Z), L.ptr(wV), L.ptr(sm), L.ptr(sl), B, n, dx, L.stream()), 'da_attn_core_fwd_bf16')
        ctx.save_for_backward(Q, K, V, E1, E2, nm_u8, ye2, wV, sm, sl)
        ctx.qscale = float(qscale)
        return Z, wV

    @staticmethod
    def backward(ctx, dZ, dwV):
        Q, K, V, E1, E2, nm_u8, ye2, wV, sm, sl = ctx.saved_tensors
        B, n, dx = Q.shape
        dZ, dwV = dZ.contiguous().to(torch.bfloat16), _c(dwV)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
        dE1, dE2 = torch.empty_like(E1), torch.empty_like(E1)
        dye1, dye2 = torch.empty_like(ye2), torch.empty_like(ye2)
        ws = _workspace(Q.device, L.lib().da_attn_core_bwd_workspace(B, n, dx))
        L.check(L.lib().da_attn_core_bwd_bf16(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm_u8), L.ptr(ye2), ctx.qscale, L.ptr(wV),
                                              L.ptr(sm), L.ptr(sl), L.ptr(dZ), L.ptr(dwV), L.ptr(dQ), L.ptr(dK), L.ptr(dV), L.ptr(dE1), L.ptr(dE2),
                                              L.ptr(dye1), L.ptr(dye2), L.ptr(ws), ws.numel(), B, n, dx, L.stream()), 'da_attn_core_bwd_bf16')
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


FUSED_WIDTHS = (16, 32, 64, 128, 256)


class ResidualLN(Function):
    """One residual block of XEyTransformerLayer (layers.py:341-359) as one kernel each way: out = LayerNorm(res + dropout_p(mask(u))),
    returned as (out fp32, out bf16 or None).  `u` fp32 or bf16; mask_mode 0 none / 1 node rows / 2 pairs (da_res_ln_fwd); dropout
    bits are a hash of (element, *seed, salt), regenerated in the backward, so no mask is stored (csrc/k_train_fused.hip)."""

    @staticmethod
    def forward(ctx, res, u, nm_u8, mask_mode, n, gamma, beta, seed, salt, p, want_bf16):
        res = _c16(res)
        ub = u.dtype == torch.bfloat16
        u = u.contiguous() if ub else _c16(u)
        D = res.shape[-1]
        s = torch.empty_like(res); out = torch.empty_like(res)
        ob = torch.empty(res.shape, dtype=torch.bfloat16, device=res.device) if want_bf16 else None
        g, be = _c(gamma), _c(beta)
        p = float(p)
        L.check(L.lib().da_res_ln_fwd(L.ptr(res), L.ptr(u), int(ub), L.ptr(nm_u8 if mask_mode else None), mask_mode, n, L.ptr(g), L.ptr(be),
                                      L.ptr(seed if p > 0 else None), salt, p, L.ptr(s), L.ptr(out), L.ptr(ob), res.numel() // D, D,
                                      L.stream()), 'da_res_ln_fwd')
        ctx.save_for_backward(s, g, nm_u8, seed)
        ctx.meta = (mask_mode, n, salt, p, ub, u.dtype)
        ctx.set_materialize_grads(False)
        return out, ob

    @staticmethod
    def backward(ctx, d_out, d_ob):
        s, g, nm_u8, seed = ctx.saved_tensors
        mask_mode, n, salt, p, ub, udt = ctx.meta
        D = s.shape[-1]
        if d_out is None and d_ob is None:
            return (None,) * 11
        d_out = None if d_out is None else _c16(d_out)
        d_ob = None if d_ob is None else d_ob.contiguous()
        ds = torch.empty_like(s)
        du = torch.empty(s.shape, dtype=udt, device=s.device) if ctx.needs_input_grad[1] else None
        dg = torch.empty(D, device=s.device); db = torch.empty(D, device=s.device)
        ws = _workspace(s.device, L.lib().da_res_ln_bwd_workspace(D))
        L.check(L.lib().da_res_ln_bwd(L.ptr(s), L.ptr(g), L.ptr(d_out), L.ptr(d_ob), L.ptr(nm_u8 if mask_mode else None), mask_mode, n,
                                      L.ptr(seed if p > 0 else None), salt, p, L.ptr(ds), L.ptr(du), int(ub), L.ptr(dg), L.ptr(db), L.ptr(ws),
                                      ws.numel(), s.numel() // D, D, L.stream()), 'da_res_ln_bwd')
        return ds, du, None, None, None, dg, db, None, None, None, None


class ReluDropout(Function):
    """dropout_p(relu(h)) IN PLACE on the output of the first Linear of a feed-forward branch (layers.py:349-357), fp32 or bf16; the
    backward needs only the result (non-zero <=> passed the ReLU and kept by the dropout)."""

    @staticmethod
    def forward(ctx, h, seed, salt, p):
        assert h.is_contiguous() and h.numel() % 4 == 0
        _relu_dropout_(h, seed, salt, p)
        ctx.mark_dirty(h)
        ctx.save_for_backward(h)
        ctx.p = float(p)
        return h

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return _relu_dropout_bwd(h, dy, ctx.p), None, None, None


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
