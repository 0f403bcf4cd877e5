"""GPU: loss values of `training_step` through the fused inference path (eval / no_grad) and d ce / d logits (da_ce_grad) against
autograd on the oracle and the reference's recorded losses (tests/golden/chain_elbo_tiny.npz, make_golden.py:432-469).  The
differentiable path and every parameter gradient: tests/test_gpu_backward.py.

Tolerances: d ce / d logits is an fp32 restatement -> 1e-6 absolute on the oracle's logits; values that go through the
bf16 HIP denoiser (loss, parameter-gradient norms) -> 3e-2 relative against the fp32 reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import ATOMS, BONDS
from tests.test_gpu_elbo import setup, close, T, KX, KE, D

pytestmark = pytest.mark.gpu


def test_ce_grad_vs_autograd_on_oracle():
    from diffalign_amd import training, graph
    z, model, data, _ = setup()
    B, n = data.X.shape[:2]
    g = torch.Generator().manual_seed(5)
    lX = (torch.randn(B, n, KX, generator=g) * 3).requires_grad_()
    lE = (torch.randn(B, n, n, KE, generator=g) * 3).requires_grad_()
    masks = graph.get_node_masks(data, list(ATOMS), list(BONDS), 'product_and_sn', 'product_and_sn')
    _, _, loss = O.ce_loss(lX, lE, data.X.cpu(), data.E.cpu(), masks[0].cpu().bool(), masks[1].cpu().bool(), 5.0)
    loss.backward()
    pred = data.get_new_object(X=lX.detach().to(D), E=lE.detach().to(D))
    gX, gE = training.ce_grad(pred, data, masks, lambda_E=5.0)
    assert (gX.cpu() - lX.grad).abs().max() <= 1e-6 and (gE.cpu() - lE.grad).abs().max() <= 1e-6
    assert gX.abs().sum() > 0 and gE.abs().sum() > 0
    # rows outside the free masks carry no gradient
    mx = masks[0].bool(); me = masks[1].bool()
    assert torch.all(gX[~mx] == 0) and torch.all(gE[~(me[:, :, None] & me[:, None, :])] == 0)
    # switches of helpers.ce: diffuse_nodes / diffuse_edges
    g0X, g0E = training.ce_grad(pred, data, masks, lambda_E=5.0, diffuse_nodes=False, diffuse_edges=False)
    assert torch.all(g0X == 0) and torch.all(g0E == 0)


def test_training_step_ce_loss_matches_reference():
    z, model, data, pos = setup()
    t = T(z['train_t']).float().to(D)
    zt = data.get_new_object(X=F.one_hot(T(z['train_zt_X']), KX).float().to(D), E=F.one_hot(T(z['train_zt_E']), KE).float().to(D), y=t)
    loss_X, loss_E, loss = model.training_step(data, 0, t_int=t, z_t=zt)
    close(loss, z['train_ce_loss'], rtol=3e-2, atol=1e-3)
    assert loss_X.dim() == 1 and loss_E.dim() == 1 and loss_X.numel() > 0 and loss_E.numel() > 0
    close(loss_X.mean() + 5 * loss_E.mean(), loss, rtol=1e-5, atol=1e-6)


def test_training_step_draws_its_own_noise_and_other_losses():
    z, model, data, pos = setup()
    torch.manual_seed(3)
    _, _, l1 = model.training_step(data, 0)
    assert torch.isfinite(l1) and float(l1) > 0
    t = T(z['train_t']).float().to(D)
    zt = data.get_new_object(X=F.one_hot(T(z['train_zt_X']), KX).float().to(D), E=F.one_hot(T(z['train_zt_E']), KE).float().to(D), y=t)
    for kind in ('vb', 'vbce'):
        model.cfg.train.loss = kind
        _, _, l = model.training_step(data, 0, t_int=t, z_t=zt.get_new_object())
        close(l, z[f'train_{kind}_loss'], rtol=3e-2, atol=1e-2)
    model.cfg.train.loss = 'ce'
