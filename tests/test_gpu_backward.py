"""GPU: the training backward -- every HIP autograd.Function against torch autograd of the same formula, the
whole denoiser's parameter gradients against autograd through the oracle (which tests/test_oracle_golden.py pins to the
gradient norms the reference's own `loss.backward()` produced), the fused AdamW against torch.optim.AdamW(amsgrad=True), and a
few optimisation steps end to end.

Tolerances (fp32 everywhere; sums over up to n*n rows are ordered differently from torch's): per-Function gradients 2e-5
relative to the largest entry; whole-model parameter gradients 1e-4 of each parameter's gradient norm (measured: 9e-6 worst);
loss values 2e-5 relative and gradient norms 5e-4 relative against the reference's recorded ones; AdamW 1e-6 absolute after 6 steps."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import weights_of, ATOMS, BONDS
from tests.test_gpu_elbo import setup, T, KX, KE, D
from tests.test_oracle_golden import oracle_training_loss, GNORM_PARAMS, _tiny_setup

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def ragged_mask(B, n, g):
    lens = torch.randint(n // 2, n + 1, (B,), generator=g)
    lens[0] = n
    return torch.arange(n)[None, :] < lens[:, None]


@pytest.mark.parametrize('B,n,dx,heads', [(3, 20, 64, 4), (2, 37, 256, 8)])
def test_attention_core_forward_and_backward(B, n, dx, heads):
    from diffalign_amd.autograd_ops import AttnCore, qscale_of
    from diffalign_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    nm = ragged_mask(B, n, g)
    xm = nm[:, :, None].float()
    mk = lambda *s: torch.randn(*s, generator=g).requires_grad_()
    Q, K, V, E1, E2, ye1, ye2 = mk(B, n, dx), mk(B, n, dx), mk(B, n, dx), mk(B, n, n, dx), mk(B, n, n, dx), mk(B, dx), mk(B, dx)
    # layers.py:225-265 restated with torch ops (oracle.node_edge_block without the projections)
    em = xm[:, :, None, :] * xm[:, None, :, :]
    Qm, Km, Vm = Q * xm, K * xm, V * xm
    Y = Qm[:, :, None, :] * Km[:, None, :, :] / math.sqrt(dx // heads) * (E1 * em + 1) + E2 * em
    Z_ref = ye1[:, None, None, :] + (ye2[:, None, None, :] + 1) * Y
    wV_ref = (F.softmax(Y, dim=2) * Vm[:, None, :, :]).sum(2)
    gz, gw = torch.randn(B, n, n, dx, generator=g), torch.randn(B, n, dx, generator=g)
    ((Z_ref * gz).sum() + (wV_ref * gw).sum()).backward()
    want = [t.grad.clone() for t in (Q, K, V, E1, E2, ye1, ye2)]
    dev = [t.detach().to(D).requires_grad_() for t in (Q, K, V, E1, E2, ye1, ye2)]
    xmd = xm.to(D)
    Z, wV = AttnCore.apply(dev[0] * xmd, dev[1] * xmd, dev[2] * xmd, dev[3], dev[4], L.as_u8(nm.to(D)), dev[5], dev[6], qscale_of(dx, heads))
    assert rel(Z, Z_ref) <= 1e-5 and rel(wV, wV_ref) <= 1e-5
    ((Z * gz.to(D)).sum() + (wV * gw.to(D)).sum()).backward()
    for name, t, w in zip(('Q', 'K', 'V', 'E1', 'E2', 'ye1', 'ye2'), dev, want):
        assert rel(t.grad, w) <= 2e-5, (name, rel(t.grad, w))
    # pairs with a padded end carry no gradient into the edge projections
    dead = ~(nm[:, :, None] & nm[:, None, :])
    assert torch.all(dev[3].grad.cpu()[dead] == 0) and torch.all(dev[4].grad.cpu()[dead] == 0)


@pytest.mark.parametrize('Dw', [32, 64, 128, 256])
def test_add_layernorm_backward(Dw):
    from diffalign_amd.autograd_ops import AddLayerNorm
    g = torch.Generator().manual_seed(Dw)
    a, b = torch.randn(5, 33, Dw, generator=g).requires_grad_(), torch.randn(5, 33, Dw, generator=g).requires_grad_()
    gam, bet = (torch.rand(Dw, generator=g) + 0.5).requires_grad_(), torch.randn(Dw, generator=g).requires_grad_()
    gy = torch.randn(5, 33, Dw, generator=g)
    ref = F.layer_norm(a + b, (Dw,), gam, bet, 1e-5)
    (ref * gy).sum().backward()
    dev = [t.detach().to(D).requires_grad_() for t in (a, b, gam, bet)]
    out = AddLayerNorm.apply(*dev)
    assert rel(out, ref) <= 1e-5
    (out * gy.to(D)).sum().backward()
    for name, t, w in zip('a b gamma beta'.split(), dev, (a, b, gam, bet)):
        assert rel(t.grad, w.grad) <= 2e-5, (name, rel(t.grad, w.grad))


@pytest.mark.parametrize('Dw,mode,ub', [(64, 2, True), (64, 2, False), (256, 1, False), (64, 0, False), (16, 0, True), (128, 2, True), (32, 1, False)])
def test_fused_residual_block_without_dropout_equals_torch(Dw, mode, ub):
    """da_res_ln_fwd / _bwd at p = 0 (csrc/k_train_fused.hip) == LayerNorm(res + mask(u)) under torch autograd: both outputs (fp32 and
    the bf16 copy), both incoming gradients summed, node-row and pair masks, fp32 and bf16 branch storage; ragged row counts."""
    from diffalign_amd.autograd_ops import ResidualLN
    g = torch.Generator().manual_seed(Dw + mode)
    B, n = 3, 7
    shape = {0: (B, Dw), 1: (B, n, Dw), 2: (B, n, n, Dw)}[mode]
    nm = torch.ones(B, n, dtype=torch.bool); nm[0, 5:] = False; nm[2, 2:] = False
    res = torch.randn(*shape, generator=g)
    u = torch.randn(*shape, generator=g)
    if ub: u = u.to(torch.bfloat16).float()
    gam, bet = torch.rand(Dw, generator=g) + 0.5, torch.randn(Dw, generator=g)
    gy, gyb = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g).to(torch.bfloat16).float()
    mask = {0: torch.ones(()), 1: nm[:, :, None].float(), 2: (nm[:, :, None] & nm[:, None, :])[..., None].float()}[mode]
    ref_in = [t.clone().double().requires_grad_() for t in (res, u, gam, bet)]
    ref = F.layer_norm(ref_in[0] + ref_in[1] * mask.double(), (Dw,), ref_in[2], ref_in[3], 1e-5)
    (ref * (gy + gyb).double()).sum().backward()
    dev = [res.to(D).requires_grad_(), (u.to(torch.bfloat16) if ub else u).to(D).requires_grad_(), gam.to(D).requires_grad_(), bet.to(D).requires_grad_()]
    out, ob = ResidualLN.apply(dev[0], dev[1], nm.to(D).to(torch.uint8), mode, n, dev[2], dev[3], None, 3, 0.0, True)
    assert rel(out, ref.float()) <= 1e-5 and ob.dtype == torch.bfloat16 and rel(ob.float(), ref.float()) <= 1e-2
    ((out * gy.to(D)).sum() + (ob.float() * gyb.to(D)).sum()).backward()
    for name, t, w in zip('res u gamma beta'.split(), dev, ref_in):
        tol = 2e-5 if not (ub and name == 'u') else 1e-2
        assert rel(t.grad.float(), w.grad.float()) <= tol, (name, rel(t.grad.float(), w.grad.float()))
    # only one of the two outputs used downstream: the other gradient arrives as None
    dev2 = [t.detach().clone().requires_grad_() for t in dev]
    out2, _ = ResidualLN.apply(dev2[0], dev2[1], nm.to(D).to(torch.uint8), mode, n, dev2[2], dev2[3], None, 3, 0.0, False)
    (out2 * (gy + gyb).to(D)).sum().backward()
    assert rel(dev2[0].grad, ref_in[0].grad.float()) <= 2e-5


def test_fused_dropout_masks_are_regenerated_in_the_backward():
    """p = 0.1: the keep rate is 0.9 within sampling error, kept elements are scaled by 65536 / (65536 - round(p 65536)), the same
    (seed, salt) gives the same mask and another salt or seed another one, and the backward applies exactly the forward's mask."""
    from diffalign_amd.autograd_ops import ResidualLN, ReluDropout
    g = torch.Generator().manual_seed(12)
    B, n, Dw, p = 4, 40, 64, 0.1
    nm = torch.ones(B, n, dtype=torch.uint8, device=D)
    res = torch.zeros(B, n, n, Dw, device=D)
    u = (torch.rand(B, n, n, Dw, generator=g) + 0.5).to(D).requires_grad_()
    seed = torch.tensor([12345], dtype=torch.int64, device=D)
    gam, bet = torch.ones(Dw, device=D), torch.zeros(Dw, device=D)
    scale = 65536.0 / (65536 - round(p * 65536))

    def run(seed, salt):
        uu = u.detach().clone().requires_grad_()
        r = res.clone().requires_grad_()
        out, _ = ResidualLN.apply(r, uu, nm, 2, n, gam, bet, seed, salt, p, False)
        gy = torch.randn(out.shape, device=D, generator=torch.Generator(device=D).manual_seed(3))
        out.backward(gy)
        s = F.layer_norm  # noqa: F841  (s itself is internal; recover the mask from the gradient pattern)
        keep = uu.grad != 0
        return keep, uu.grad, r.grad

    keep, du, dres = run(seed, 5)
    rate = float(keep.float().mean())
    assert abs(rate - 0.9) < 3e-3, rate
    assert torch.equal(du[keep], (dres * scale)[keep]) and torch.all(du[~keep] == 0)          # d u = mask * scale * d s
    keep_again, _, _ = run(seed, 5)
    assert torch.equal(keep, keep_again)
    keep_salt, _, _ = run(seed, 6)
    keep_seed, _, _ = run(seed + 1, 5)
    for other in (keep_salt, keep_seed):                               # independent masks agree on 0.81 + 0.01 of the elements
        assert abs(float((keep == other).float().mean()) - 0.82) < 5e-3
    # rows see different masks; per-channel keep rates are all close to 0.9
    assert float(keep.float().mean(dim=(0, 1, 2)).min()) > 0.88 and float(keep.float().mean(dim=(0, 1, 2)).max()) < 0.92
    for dt in (torch.float32, torch.bfloat16):
        h = torch.randn(3, 50, 128, generator=g).to(D).to(dt).requires_grad_()
        o = ReluDropout.apply(h * 1.0, seed, 9, p)
        pos = h.detach() > 0
        kept = o.detach() != 0
        assert torch.all(kept <= pos) and abs(float(kept[pos].float().mean()) - 0.9) < 2e-2
        assert rel(o.detach()[kept].float(), (h.detach().float() * scale)[kept]) <= (1e-6 if dt == torch.float32 else 8e-3)
        gy = torch.randn(3, 50, 128, generator=g).to(D).to(dt)
        o.backward(gy)
        assert rel(h.grad.float(), (gy.float() * kept * scale)) <= (1e-6 if dt == torch.float32 else 8e-3)
        o0 = ReluDropout.apply(h.detach() * 1.0, None, 9, 0.0)
        assert torch.equal(o0, torch.relu(h.detach()))


def test_layernorm_backward_rejects_unbuilt_width():
    from diffalign_amd import _lib as L
    s = torch.randn(4, 48, device=D)
    ws = torch.empty(8 * 1024 * 48, dtype=torch.uint8, device=D)
    rc = L.lib().da_layernorm_bwd(L.ptr(s), L.ptr(torch.ones(48, device=D)), L.ptr(s), L.ptr(torch.empty_like(s)), L.ptr(torch.empty(48, device=D)),
                                  L.ptr(torch.empty(48, device=D)), L.ptr(ws), ws.numel(), 4, 48, L.stream())
    assert rc != 0 and 'width 48' in L.last_error()


@pytest.mark.parametrize('shape', [(3, 17, 64), (2, 9, 9, 32), (2, 1, 8)])
def test_pooled_statistics_forward_and_backward(shape):
    from diffalign_amd.autograd_ops import PoolStats
    g = torch.Generator().manual_seed(5)
    Tt = torch.randn(*shape, generator=g).requires_grad_()
    dims = tuple(range(1, len(shape) - 1))
    N = int(np.prod(shape[1:-1]))
    ref = O._stats4(Tt, dims) if N > 1 else torch.cat([Tt.mean(dims), Tt.amin(dims), Tt.amax(dims), torch.zeros(shape[0], shape[-1])], -1)
    gz = torch.randn(shape[0], 4 * shape[-1], generator=g)
    (ref * gz).sum().backward()
    td = Tt.detach().to(D).requires_grad_()
    out = PoolStats.apply(td)
    if N > 1:
        assert rel(out, ref) <= 1e-5
    else:   # torch.std of one row is nan (the reference never pools a single row); the kernel returns 0 there
        assert rel(out[:, :3 * shape[-1]], ref[:, :3 * shape[-1]]) <= 1e-6 and torch.all(out[:, 3 * shape[-1]:] == 0)
    (out * gz.to(D)).sum().backward()
    assert rel(td.grad, Tt.grad) <= 2e-5


@pytest.mark.parametrize('xb,yb', [(True, True), (True, False), (False, True)])
def test_wide_linear_weight_gradient_bf16_storage(xb, yb):
    """da_linear_wgrad_mixed: bf16 operands are widened on load (one of them bf16: fp32 MFMA) or fed to the bf16 MFMA through an exact
    in-register transposition (both bf16); products and the reduction are fp32 -> equal to the fp32 kernel on the same
    (bf16-representable) values up to the order of the partial sums.  Ragged row counts incl. fewer than one 32-row stage."""
    from diffalign_amd import _lib as L
    from diffalign_amd.autograd_ops import _workspace
    g = torch.Generator().manual_seed(9)
    for fin, fout, R in ((64, 256, 5000), (256, 64, 777), (64, 128, 4097), (128, 64, 31), (64, 64, 70001), (128, 128, 1)):
        x = torch.randn(R, fin, generator=g).to(torch.bfloat16).to(D); dy = torch.randn(R, fout, generator=g).to(torch.bfloat16).to(D)
        xa, ya = (x if xb else x.float()), (dy if yb else dy.float())
        dW = torch.empty(fout, fin, device=D); db = torch.empty(fout, device=D)
        ws = _workspace(torch.device(D), L.lib().da_linear_wgrad_workspace(R, fin, fout))
        L.check(L.lib().da_linear_wgrad_mixed(L.ptr(xa), int(xb), L.ptr(ya), int(yb), L.ptr(dW), L.ptr(db), L.ptr(ws), ws.numel(), R, fin, fout,
                                              L.stream()), 'wgrad_mixed')
        want = dy.double().t() @ x.double()
        assert float((dW.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
        assert float((db.double() - dy.double().sum(0)).abs().max()) <= 1e-4 * float(dy.double().sum(0).abs().max())


@pytest.mark.parametrize('fin,fout', [(64, 256), (256, 64), (64, 128), (128, 64), (64, 64), (128, 128)])
def test_wide_linear_weight_gradient(fin, fout):
    """da_linear_wgrad (split over rows, fp32 MFMA, ordered fold) == torch's dY^T X and column sums; ragged row counts."""
    from diffalign_amd.autograd_ops import WideLinear
    from diffalign_amd import _lib as L
    g = torch.Generator().manual_seed(fin + fout)
    for R in (1, 63, 1000, 70001):
        x, w, b = torch.randn(R, fin, generator=g), torch.randn(fout, fin, generator=g) * 0.1, torch.randn(fout, generator=g)
        gy = torch.randn(R, fout, generator=g)
        xd, wd, bd = (t.to(D).requires_grad_() for t in (x, w, b))
        out = WideLinear.apply(xd, wd, bd)
        out.backward(gy.to(D))
        xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
        F.linear(xr, wr, br).backward(gy.double())
        assert rel(out, F.linear(x, w, b)) <= 1e-5
        assert rel(wd.grad, wr.grad) <= 5e-6 and rel(bd.grad, br.grad) <= 5e-6 and rel(xd.grad, xr.grad) <= 1e-5, (R, rel(wd.grad, wr.grad))
    rc = L.lib().da_linear_wgrad(L.ptr(xd.detach()), L.ptr(xd.detach()), L.ptr(wd.detach()), None, L.ptr(xd.detach()), 1 << 40, 8, 48, 48, L.stream())
    assert rc != 0 and 'not built' in L.last_error()


@pytest.mark.parametrize('precision,fused', [('fp32', True), ('bf16', True), ('fp32', False), ('bf16', False)])
def test_full_width_layers_gradients_vs_oracle(precision, fused):
    """(bf16: the mixed-precision training path -- bf16 storage and GEMM operands for the per-pair tensors, fp32 accumulation and master
    weights, as the reference's fp16 autocast -- against the same fp32 oracle at a bf16 tolerance: logits 3e-2 of their range, every
    parameter gradient 5e-2 relative.)
    align_absorbing widths (dx 256, de 64, dy 64, 8 heads), 2 layers, ragged B=3 x n=26: every parameter's gradient of a random
    linear functional of the logits, training forward + HIP backward vs autograd through the oracle (fp32 CPU).  The per-pair Linears
    take the WideLinear path here (the tiny golden model's widths do not)."""
    import torch.nn as nn
    from bench import FULL
    from diffalign_amd import autograd_ops, synthetic as S
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
    torch.manual_seed(4)
    B, n = 3, 26
    m = GraphTransformerWithYAtomMapPosEmb(2, {'X': KX + 8, 'E': KE, 'y': 13}, FULL['hidden_mlp_dims'], FULL['hidden_dims'],
                                           {'X': KX, 'E': KE, 'y': 1}, nn.ReLU(), nn.ReLU(), dropout=0.0, p_to_r_skip_connection=True,
                                           p_to_r_init=1.0, alignment_type='correct')
    true = S.make_reaction_batch(B, n, seed=5, P=None)
    g = torch.Generator().manual_seed(6)
    X = torch.cat([true['X'], torch.rand(B, n, 8, generator=g)], -1)
    y = torch.rand(B, 13, generator=g)
    pos = S.shared_pos_encodings(true, 256, seed=7)
    gX, gE = torch.randn(B, n, KX, generator=g), torch.randn(B, n, n, KE, generator=g)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    lX, lE, _ = O.denoiser_forward(sd, X, true['E'], y, true['node_mask'], 2, 8, (KX, KE, 1), true['atom_map_numbers'], pos,
                                   true['mol_assignments'], True)
    ((lX * gX).sum() + (lE * gE).sum()).backward()
    m = m.to(D).train()
    m.train_precision = precision
    m.train_fused = fused              # the residual blocks as fused kernels (csrc/k_train_fused.hip) or as torch elementwise passes
    tol_logit = 2e-4 if precision == 'fp32' else 3e-2 * float(lE.detach().abs().max())
    tol_grad = 2e-4 if precision == "fp32" else 1e-1
    old = autograd_ops.WIDE_MIN_ROWS
    autograd_ops.WIDE_MIN_ROWS = 1
    try:
        Xo, Eo, _, _ = m(X.to(D), true['E'].to(D), y.to(D), true['node_mask'].to(D), true['atom_map_numbers'].to(D), pos.to(D),
                         true['mol_assignments'].to(D))
        assert (Xo.detach().cpu() - lX.detach()).abs().max() <= tol_logit and (Eo.detach().cpu() - lE.detach()).abs().max() <= tol_logit
        ((Xo * gX.to(D)).sum() + (Eo * gE.to(D)).sum()).backward()
    finally:
        autograd_ops.WIDE_MIN_ROWS = old
    worst = ('', 0.0)
    for name, p in m.named_parameters():
        want = sd[name].grad
        if want is None:
            assert p.grad is None, name
            continue
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        if err > worst[1]: worst = (name, err)
        assert err <= tol_grad, (name, err)
    print(f'worst full-width parameter-gradient error ({precision})', worst)


@pytest.mark.parametrize('de', [8, 64])       # 64: the hidden width, which takes the 16-byte-per-lane form of the mask kernel
def test_hidden_mask_gradient(de):
    from diffalign_amd.autograd_ops import MaskDense
    from diffalign_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    nm = ragged_mask(3, 12, g)
    X, E = torch.randn(3, 12, 16, generator=g).requires_grad_(), torch.randn(3, 12, 12, de, generator=g).requires_grad_()
    rX, rE = O.mask_dense(X, E, nm)
    gX, gE = torch.randn(3, 12, 16, generator=g), torch.randn(3, 12, 12, de, generator=g)
    ((rX * gX).sum() + (rE * gE).sum()).backward()
    Xd, Ed = X.detach().to(D).requires_grad_(), E.detach().to(D).requires_grad_()
    oX, oE = MaskDense.apply(Xd, Ed, L.as_u8(nm.to(D)))
    assert torch.equal(oX.cpu(), rX.detach()) and torch.equal(oE.cpu(), rE.detach())
    ((oX * gX.to(D)).sum() + (oE * gE.to(D)).sum()).backward()
    assert torch.equal(Xd.grad.cpu(), X.grad) and torch.equal(Ed.grad.cpu(), E.grad)


def _train_inputs(z, data):
    t = T(z['train_t']).float().to(D)
    zt = data.get_new_object(X=F.one_hot(T(z['train_zt_X']), KX).float().to(D), E=F.one_hot(T(z['train_zt_E']), KE).float().to(D), y=t)
    return t, zt


def test_training_forward_matches_oracle_logits():
    """The fp32 training forward (train_forward.py) == the oracle's denoiser on the same weights: 1e-4 absolute on logits of
    magnitude ~10 (the fused bf16 inference path is held to 3e-2 relative on the same comparison)."""
    z, model, data, pos = setup()
    model.train()
    t, zt = _train_inputs(z, data)
    pred = model.forward(z_t=zt)
    assert pred.X.requires_grad and pred.E.requires_grad
    sd = {k: v.float() for k, v in weights_of(z).items()}
    lX, lE, _ = O.denoiser_forward(sd, zt.X.cpu(), zt.E.cpu(), t.cpu(), T(z['node_mask']), 2, int(z['n_head']), (KX, KE, 1), T(z['am']), pos.cpu(),
                                   T(z['mol']), True)
    assert (pred.X.detach().cpu() - lX).abs().max() <= 1e-4 and (pred.E.detach().cpu() - lE).abs().max() <= 1e-4
    model.eval()
    with torch.no_grad():
        pe = model.forward(z_t=zt)
    assert not pe.X.requires_grad
    assert (pe.X - pred.X.detach()).abs().max() <= 3e-2 * pred.X.detach().abs().max()


def test_mixed_precision_training_step_close_to_fp32():
    """The tiny golden model trained in mixed precision (module.train_precision = 'bf16'): loss within 2e-2 of the reference's fp32
    value, every parameter gradient within 1.5e-1 (relative, whole tensor; 8.5e-2 measured at linE2.weight of layer 0, whose gradient is a sum over all pairs of bf16-stored terms that largely cancel) of autograd through the fp32 oracle."""
    z, model, data, pos = setup()
    model.cfg.train.loss = 'vb'
    model.train()
    model.model.train_precision = 'bf16'
    t, zt = _train_inputs(z, data)
    _, _, loss = model.training_step(data, 0, t_int=t, z_t=zt)
    loss.backward()
    want_loss = float(z['train_vb_loss'])
    assert abs(float(loss.detach()) - want_loss) <= 2e-2 * abs(want_loss)
    _, sd, true, _ = _tiny_setup()
    sd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    oloss, _ = oracle_training_loss(z, sd, true, 'vb', pos=pos.cpu())
    oloss.backward()
    worst = ('', 0.0)
    for name, p in model.model.named_parameters():
        want = sd[name].grad
        if want is None:
            continue
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        if err > worst[1]: worst = (name, err)
        assert err <= 1.5e-1, (name, err)
    print(f'mixed precision: loss {float(loss):.5f} (fp32 reference {want_loss:.5f}); worst parameter-gradient error {worst[1]:.2e} at {worst[0]}')
    model.cfg.train.loss = 'ce'
    model.model.train_precision = None


@pytest.mark.parametrize('kind', ['ce', 'vb', 'vbce'])
def test_parameter_gradients_match_oracle_autograd_and_reference_norms(kind):
    z, model, data, pos = setup()
    model.cfg.train.loss = kind
    model.train()
    t, zt = _train_inputs(z, data)
    _, _, loss = model.training_step(data, 0, t_int=t, z_t=zt)
    assert loss.requires_grad
    loss.backward()
    want_loss = float(z[f'train_{kind}_loss'])
    assert abs(float(loss.detach()) - want_loss) <= 2e-5 * abs(want_loss) + 1e-5
    _, sd, true, _ = _tiny_setup()
    sd = {k: v.clone().float().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    oloss, _ = oracle_training_loss(z, sd, true, kind, pos=pos.cpu())
    oloss.backward()
    worst = ('', 0.0)
    for name, p in model.model.named_parameters():
        want = sd[name].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        if err > worst[1]: worst = (name, err)
        assert err <= 1e-4, (name, err, float(want.norm()))
    for pn in GNORM_PARAMS:
        want, got = float(z[f'train_{kind}_gnorm::{pn}']), float(dict(model.model.named_parameters())[pn].grad.norm())
        assert abs(got - want) <= 5e-4 * abs(want) + 1e-6, (pn, got, want)
    print(f'[{kind}] loss {float(loss):.6f} (reference {want_loss:.6f}); worst parameter-gradient error {worst[1]:.2e} at {worst[0]}')
    model.cfg.train.loss = 'ce'


def test_vb_gradient_on_logits_vs_oracle():
    """da_vb_grad alone: d elbo_batch_quick / d logits on random logits (every graph's t, one of them t == 1)."""
    from diffalign_amd.autograd_ops import DiffusionLoss
    from diffalign_amd import graph
    z, model, data, _ = setup()
    t, zt = _train_inputs(z, data)
    assert int((t == 1).sum()) == 1
    B, n = data.X.shape[:2]
    g = torch.Generator().manual_seed(21)
    lX, lE = (torch.randn(B, n, KX, generator=g) * 2).requires_grad_(), (torch.randn(B, n, n, KE, generator=g) * 2).requires_grad_()
    masks = graph.get_node_masks(data, list(ATOMS), list(BONDS), 'product_and_sn', 'product_and_sn')
    tabs = O.absorbing_tables(10, KX, KE, 1, 27, 0, [28, 0], [4, 5, 6])
    want = O.elbo_batch_quick(data.X.cpu(), data.E.cpu(), zt.X.cpu(), zt.E.cpu(), lX, lE, tabs, t.cpu()[:, 0], 10, masks[0].cpu().bool(),
                              masks[1].cpu().bool(), 5.0)
    want.backward()
    dX, dE = lX.detach().to(D).requires_grad_(), lE.detach().to(D).requires_grad_()
    got = DiffusionLoss.apply(dX, dE, model, data, zt, masks, 'vb')
    assert abs(float(got) - float(want)) <= 1e-4 * abs(float(want))
    got.backward()
    assert rel(dX.grad, lX.grad) <= 1e-4 and rel(dE.grad, lE.grad) <= 1e-4


def test_fused_adamw_matches_torch():
    from diffalign_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(2)
    shapes = [(7, 5), (13,), (1,), (3, 4, 2)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    mine = [torch.nn.Parameter(p.clone().to(D)) for p in ps]
    kw = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=True)
    o_ref, o = torch.optim.AdamW(ref, **kw), FlatAdamW(mine, **kw)
    for step in range(6):
        o_ref.zero_grad(); o.zero_grad()
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g) * (10.0 if step == 1 else 1.0)     # a spike, so that amsgrad's running max matters
            r.grad = gr.clone()
            m.grad = gr.to(D)                                                         # what backward() leaves
        o.gather()
        assert all(m.grad.data_ptr() >= o.flat_grad.data_ptr() for m in mine)         # .grad now views the flat buffer
        o_ref.step(); o.step()
    for r, m in zip(ref, mine):
        assert (m.detach().cpu() - r.detach()).abs().max() <= 1e-6
    assert all(m.data_ptr() >= o.flat_param.data_ptr() for m in mine)                 # parameters live in the flat buffer


@pytest.mark.parametrize('device_step', [False, True])
def test_adamw_checkpoint_exchange_with_torch_and_scheduler(device_step):
    """optimizer_state_dict in torch.optim.AdamW's format, both ways (training/helpers.py:99-110, utils/setup.py:331-346), with a
    parameter that never receives a gradient (torch keeps no state for it), and a torch learning-rate scheduler driving the step size."""
    from diffalign_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(7, 5), (13,), (2, 2), (3, 4, 2)]
    dead = 2                                                                            # index of the parameter without a gradient
    ps = [torch.randn(*s, generator=g) for s in shapes]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=True)
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(9)]

    def run(opt, params, lo, hi, sched=None):
        for t in range(lo, hi):
            opt.zero_grad()
            for i, p in enumerate(params):
                if i != dead:
                    p.grad = grads[t][i].to(p.device).clone()
            if isinstance(opt, FlatAdamW):
                opt.gather()
            opt.step()
            if sched is not None:
                sched.step()

    mk_sched = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=2, gamma=0.5)
    # torch for 3 steps -> checkpoint -> this optimiser for 3 steps -> checkpoint -> torch for 3 steps; reference: torch for 9 steps
    ref = [p.clone().requires_grad_() for p in ps]
    o_ref = torch.optim.AdamW(ref, **kw); s_ref = mk_sched(o_ref)
    run(o_ref, ref, 0, 9, s_ref)
    a = [p.clone().requires_grad_() for p in ps]
    o_a = torch.optim.AdamW(a, **kw); s_a = mk_sched(o_a)
    run(o_a, a, 0, 3, s_a)
    ck1 = dict(opt=o_a.state_dict(), sched=s_a.state_dict())
    assert dead not in ck1['opt']['state']
    b = [torch.nn.Parameter(p.detach().clone().to(D)) for p in a]
    o_b = FlatAdamW(b, device_step=device_step, **kw); s_b = mk_sched(o_b)
    o_b.load_state_dict(ck1['opt']); s_b.load_state_dict(ck1['sched'])
    assert o_b.steps == 3 and abs(o_b.lr - o_a.param_groups[0]['lr']) < 1e-12
    run(o_b, b, 3, 6, s_b)
    ck2 = dict(opt=o_b.state_dict(), sched=s_b.state_dict())
    assert sorted(ck2['opt']['state']) == sorted(ck1['opt']['state']) and float(ck2['opt']['state'][0]['step']) == 6.0
    c = [p.detach().cpu().clone().requires_grad_() for p in b]
    o_c = torch.optim.AdamW(c, **kw); s_c = mk_sched(o_c)
    o_c.load_state_dict({'state': {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in ck2['opt']['state'].items()},
                         'param_groups': ck2['opt']['param_groups']})
    s_c.load_state_dict(ck2['sched'])
    run(o_c, c, 6, 9, s_c)
    for r, m in zip(ref, c):
        assert (m.detach() - r.detach()).abs().max() <= 2e-6
    assert torch.equal(c[dead].detach(), ps[dead])


def test_adamw_frozen_parameters_keep_torchs_index_space():
    """The reference builds AdamW over a parameter list that can contain frozen parameters (with use_ema: the EMA copy, utils/setup.py:356).
    They stay in the parameter group's index space, out of the flat buffers: a torch-written state dict of such a module loads here and
    one written here loads in torch; a reference list that is a superset (extra frozen entries last) loads too; a parameter that joins
    late saves its own update count."""
    from diffalign_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(4)
    shapes = [(5, 3), (4,), (6, 2), (3,), (2, 2)]
    frozen = {1, 4}
    ps = [torch.randn(*s, generator=g) for s in shapes]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=True)
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(6)]
    mk = lambda dev: [torch.nn.Parameter(p.clone().to(dev), requires_grad=i not in frozen) for i, p in enumerate(ps)]

    def run(opt, params, lo, hi, skip=()):
        for t in range(lo, hi):
            opt.zero_grad()
            for i, p in enumerate(params):
                if p.requires_grad and not (i in skip and t == 0):
                    p.grad = grads[t][i].to(p.device).clone()
            if isinstance(opt, FlatAdamW):
                opt.gather()
            opt.step()

    ref = mk('cpu'); o_ref = torch.optim.AdamW(ref, **kw); run(o_ref, ref, 0, 6)
    a = mk('cpu'); o_a = torch.optim.AdamW(a, **kw); run(o_a, a, 0, 3)
    sd_a = o_a.state_dict()
    assert sorted(sd_a['state']) == [0, 2, 3] and sd_a['param_groups'][0]['params'] == [0, 1, 2, 3, 4]
    b = [torch.nn.Parameter(p.detach().clone().to(D), requires_grad=p.requires_grad) for p in a]
    o_b = FlatAdamW(b, **kw)
    assert len(o_b.params) == 3 and o_b.group_index == [0, 2, 3]
    o_b.load_state_dict(sd_a)
    run(o_b, b, 3, 6)
    sd_b = o_b.state_dict()
    assert sorted(sd_b['state']) == [0, 2, 3] and sd_b['param_groups'][0]['params'] == [0, 1, 2, 3, 4]
    c = [torch.nn.Parameter(p.detach().cpu().clone(), requires_grad=p.requires_grad) for p in b]
    o_c = torch.optim.AdamW(c, **kw)
    o_c.load_state_dict({'state': {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd_b['state'].items()},
                         'param_groups': sd_b['param_groups']})                      # torch's own group-size check passes
    for r, m in zip(ref, c):
        assert (m.detach() - r.detach()).abs().max() <= 2e-6
    # a reference checkpoint whose list has extra frozen parameters at the end (the EMA copy): matched by position
    sup = {'state': sd_a['state'], 'param_groups': [dict(sd_a['param_groups'][0], params=list(range(9)))]}
    o_d = FlatAdamW([torch.nn.Parameter(p.detach().clone().to(D), requires_grad=p.requires_grad) for p in a], **kw)
    o_d.load_state_dict(sup)
    assert o_d.steps == 3 and torch.equal(o_d.exp_avg, o_b.exp_avg * 0 + o_d.exp_avg)
    with pytest.raises(AssertionError):
        o_d.load_state_dict({'state': {7: sd_a['state'][0]}, 'param_groups': sup['param_groups']})
    # a parameter whose first gradient comes one step late saves its own count, as torch does
    e = mk(D); o_e = FlatAdamW(e, **kw); run(o_e, e, 0, 3, skip={2})
    steps = {k: float(v['step']) for k, v in o_e.state_dict()['state'].items()}
    assert steps == {0: 3.0, 2: 2.0, 3: 3.0}


def test_optimisation_steps_reduce_the_loss_and_refresh_the_inference_weights():
    """train_one_step x 8 on one fixed batch (ce): the loss falls; with dropout > 0 the step stays finite; the fused bf16
    inference path re-packs the updated weights (its loss follows the trained fp32 one)."""
    from diffalign_amd.optim import FlatAdamW
    z, model, data, pos = setup()
    model.train()
    t, zt = _train_inputs(z, data)
    opt = FlatAdamW(model.model.parameters(), lr=2e-3)
    losses = [float(model.train_one_step(data, opt, t_int=t, z_t=zt.get_new_object())) for _ in range(8)]
    assert all(math.isfinite(l) for l in losses) and losses[-1] < 0.8 * losses[0], losses
    model.eval()
    with torch.no_grad():
        _, _, l_eval = model.training_step(data, 0, t_int=t, z_t=zt.get_new_object())
    assert abs(float(l_eval) - losses[-1]) <= 0.15 * losses[0]
    model.train()
    for layer in model.model.tf_layers:
        layer.dropout_p = 0.1
    torch.manual_seed(0)
    l_drop = float(model.train_one_step(data, opt, t_int=t, z_t=zt.get_new_object()))
    assert math.isfinite(l_drop)
    print('losses', [round(l, 4) for l in losses], 'eval', float(l_eval), 'with dropout', l_drop)


def test_device_step_adamw_and_graphed_training_step():
    """optim.FlatAdamW(device_step=True) (bias corrections from a step counter in device memory) == the host-step form, and a whole
    training step replayed from a HIP graph (graphs.GraphedTrainStep) trains: the loss on a fixed (t, z_t) probe falls, the counter
    advances once per replay, the bf16 inference weights are re-packed."""
    from diffalign_amd.optim import FlatAdamW
    from diffalign_amd.graphs import GraphedTrainStep
    g = torch.Generator().manual_seed(8)
    ps = [torch.randn(33, 7, generator=g), torch.randn(5, generator=g)]
    a = [torch.nn.Parameter(p.clone().to(D)) for p in ps]; b = [torch.nn.Parameter(p.clone().to(D)) for p in ps]
    oa, ob = FlatAdamW(a, lr=1e-3, weight_decay=1e-2), FlatAdamW(b, lr=1e-3, weight_decay=1e-2, device_step=True)
    for _ in range(5):
        for o, params in ((oa, a), (ob, b)):
            o.zero_grad()
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(D)
            x.grad = gr.clone(); y.grad = gr.clone()
        for o in (oa, ob):
            o.gather(); o.step()
    assert int(ob.step_dev) == 5
    for x, y in zip(a, b):
        assert (x - y).abs().max() <= 1e-7

    z, model, data, pos = setup()
    model.train()
    t, zt = _train_inputs(z, data)
    probe = lambda: float(model.training_step(data, 0, t_int=t, z_t=zt.get_new_object())[2])
    with torch.no_grad():
        model.eval(); l0 = probe(); model.train()
    opt = FlatAdamW(model.model.parameters(), lr=2e-3, device_step=True)
    torch.manual_seed(1)
    w_before = opt.flat_param.clone()
    gstep = GraphedTrainStep(model, opt, data, warmup=2)
    n0 = int(opt.step_dev)
    # building the graph does not train: weights, moments and both step counters are what they were (warm-up steps taken back)
    assert n0 == 0 and opt.steps == 0 and torch.equal(opt.flat_param, w_before) and float(opt.exp_avg.abs().max()) == 0.0
    losses = [float(gstep(data)) for _ in range(12)]
    assert int(opt.step_dev) == n0 + 12 and all(math.isfinite(l) for l in losses)
    assert opt.steps == int(opt.step_dev)                                            # what state_dict() saves == what the kernel used
    assert all(float(v['step']) == 12.0 for v in opt.state_dict()['state'].values())
    with torch.no_grad():
        model.eval(); l1 = probe()
    assert l1 < 0.8 * l0, (l0, l1)
    print('graphed training: probe loss', l0, '->', l1)
    # positional encodings of the next batch computed beside the current replay (next_data=): the step that follows uses exactly them
    model.train()
    model.cfg.neuralnet.pos_encoding_type = 'laplacian_pos_enc'; model.cfg.neuralnet.num_lap_eig_vectors = 8
    saved_fn, model.pos_encodings_fn = getattr(model, 'pos_encodings_fn', None), None
    g2 = GraphedTrainStep(model, opt, data, warmup=1)
    assert g2.pos is not None
    other = data.get_new_object()
    other.E = other.E.flip(0).contiguous(); other.X = other.X.flip(0).contiguous()
    other.atom_map_numbers = other.atom_map_numbers.flip(0).contiguous(); other.mol_assignments = other.mol_assignments.flip(0).contiguous()
    other.node_mask = other.node_mask.flip(0).contiguous()
    g2(data, next_data=other)
    assert g2._ahead[0] is other
    l_a = float(g2(other))
    assert '_ahead' not in g2.__dict__ and math.isfinite(l_a)
    want = model.get_pos_encodings(other)
    # (the Laplacian encodings carry random signs per call: compare up to the sign of each eigenvector column)
    assert torch.allclose(g2.pos.abs(), want.abs(), atol=1e-5)
    model.pos_encodings_fn = saved_fn


def test_plain_graph_transformer_gradients_vs_oracle():
    """GraphTransformerWithY (no atom maps, no positional encodings, no skip connection: transformer_model_with_y.py:33-112) in
    train mode: logits and every parameter gradient against autograd through the oracle."""
    import torch.nn as nn
    from tests.golden.common import tiny_dims
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithY
    dims = tiny_dims()
    torch.manual_seed(12)
    B, n = 3, 19
    m = GraphTransformerWithY(2, {'X': KX, 'E': KE, 'y': 1}, dims['hidden_mlp_dims'], dims['hidden_dims'], {'X': KX, 'E': KE, 'y': 1},
                              nn.ReLU(), nn.ReLU(), dropout=0.0)
    g = torch.Generator().manual_seed(13)
    nm = ragged_mask(B, n, g)
    X = F.one_hot(torch.randint(0, KX, (B, n), generator=g), KX).float()
    Ec = torch.triu(torch.randint(0, KE, (B, n, n), generator=g), 1); Ec = Ec + Ec.transpose(1, 2)
    E = F.one_hot(Ec, KE).float()
    y = torch.rand(B, 1, generator=g)
    gX, gE = torch.randn(B, n, KX, generator=g), torch.randn(B, n, n, KE, generator=g)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    lX, lE, _ = O.denoiser_forward(sd, X, E, y, nm, 2, dims['hidden_dims']['n_head'], (KX, KE, 1))
    ((lX * gX).sum() + (lE * gE).sum()).backward()
    m = m.to(D).train()
    Xo, Eo, yo, _ = m(X.to(D), E.to(D), y.to(D), nm.to(D))
    assert (Xo.detach().cpu() - lX.detach()).abs().max() <= 1e-4 and (Eo.detach().cpu() - lE.detach()).abs().max() <= 1e-4
    ((Xo * gX.to(D)).sum() + (Eo * gE.to(D)).sum()).backward()
    for name, p in m.named_parameters():
        want = sd[name].grad
        if want is None:
            assert p.grad is None, name
            continue
        err = float((p.grad.cpu() - want).norm()) / (float(want.norm()) + 1e-6)
        assert err <= 1e-4, (name, err)
