"""GPU: cfg.neuralnet.use_ema on the host class (diffusion_abstract.py:94-97, 333-339; training/helpers.py:143-144): forward() outside
training runs the EMA copy (fused bf16 inference path, re-packed when the average moves), train_one_step updates it after the
optimiser step, and the averaged weights follow the reference's update rule (tests/test_ema.py pins that rule to the reference's class)."""
import pytest
import torch

from tests.test_gpu_elbo import setup, T, D

pytestmark = pytest.mark.gpu


def ema_model():
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from tests.golden.common import tiny_dims, weights_of
    from tests.test_gpu_sampling import make_cfg, make_info
    z, _, data, pos = setup()
    cfg = make_cfg(tiny_dims())
    cfg.neuralnet.use_ema, cfg.neuralnet.ema_decay = True, 0.9
    model = DiscreteDenoisingDiffusion(cfg, make_info(), device=D)
    model.model.load_state_dict(weights_of(z))
    model.ema.copy_params_from_model_to_ema()
    model.pos_encodings_fn = lambda zt: pos
    model.ema.update_after_step, model.ema.update_every = 1, 1        # average from the third step on
    return z, model, data


def test_eval_forward_runs_the_averaged_weights_and_training_updates_them():
    from diffalign_amd.optim import FlatAdamW
    z, model, data = ema_model()
    t = T(z['train_t']).float().to(D)
    zt = data.get_new_object(y=t)
    model.eval()
    with torch.no_grad():
        p0 = model.forward(z_t=zt)
    model.train()
    opt = FlatAdamW(model.model.parameters(), lr=2e-3)
    online, avg = dict(model.model.named_parameters()), dict(model.ema.ema_model.named_parameters())
    expect = {k: v.detach().clone() for k, v in avg.items()}
    for step in range(6):
        model.train_one_step(data, opt, t_int=t)
        # ema_pytorch.py:133-148, 124-131 with update_after_step = 1, update_every = 1, power = 1, beta = 0.9
        if step <= 2:
            expect = {k: online[k].detach().clone() for k in expect}      # steps 0-1: copies; step 2: first averaged step starts from a copy, decay 0
        else:
            decay = min(1 - (1 + (step - 1)) ** -1.0, 0.9)
            expect = {k: expect[k] + (1 - decay) * (online[k].detach() - expect[k]) for k in expect}
    for k in expect:
        assert torch.allclose(avg[k], expect[k], rtol=1e-5, atol=1e-7), k
    assert not torch.equal(avg['mlp_in_X.0.weight'], online['mlp_in_X.0.weight'])
    model.eval()
    with torch.no_grad():
        p1 = model.forward(z_t=zt)                  # EMA copy, re-packed
        model.ema.ema_model.load_state_dict(model.model.state_dict())
        p2 = model.forward(z_t=zt)                  # EMA copy == online weights
    assert not torch.equal(p0.X, p1.X) and not torch.equal(p1.X, p2.X)
    ema, model.ema = model.ema, None
    with torch.no_grad():
        p3 = model.forward(z_t=zt)                  # the online model itself
    model.ema = ema
    assert torch.equal(p2.X, p3.X) and torch.equal(p2.E, p3.E)
    # sampling uses the averaged weights too (fused reverse step on the EMA copy)
    out = model.sample_one_batch(data=data, chain_seeds=torch.arange(4))
    assert bool((out.X.sum(-1) == 1).all()) and torch.equal(out.E, out.E.transpose(1, 2))


def test_graphed_training_step_updates_ema_outside_the_graph():
    from diffalign_amd.optim import FlatAdamW
    from diffalign_amd.graphs import GraphedTrainStep
    z, model, data = ema_model()
    model.train()
    opt = FlatAdamW(model.model.parameters(), lr=2e-3, device_step=True)
    step = GraphedTrainStep(model, opt, data, warmup=2)
    assert int(model.ema.step) == 0                  # warm-up and capture left the average alone
    for _ in range(5):
        step(data)
    torch.cuda.synchronize()
    assert int(model.ema.step) == 5
    online, avg = dict(model.model.named_parameters()), dict(model.ema.ema_model.named_parameters())
    d = (avg['mlp_in_X.0.weight'] - online['mlp_in_X.0.weight'].detach()).abs().max()
    assert 0 < float(d) < 0.1


def test_checkpoint_resume_reproduces_the_run(tmp_path):
    """training/helpers.py:99-110 + utils/setup.py:331-346: save after 3 steps, load into fresh objects, 3 more steps on both: same
    weights, optimiser state, learning rate and EMA (the backward's float atomics leave 1e-6 of run-to-run difference)."""
    from diffalign_amd.optim import FlatAdamW
    from diffalign_amd import checkpoint as CK
    z, model, data = ema_model()
    t = T(z['train_t']).float().to(D)
    model.train()
    opt = FlatAdamW(model.model.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)

    def run(m, o, s, k):
        for _ in range(k):
            m.train_one_step(data, o, t_int=t, z_t=None, noise=None)
            s.step()

    torch.manual_seed(1)
    run(model, opt, sched, 3)
    path = str(tmp_path / 'epoch3.pt')
    CK.save_checkpoint(path, model, opt, sched)
    ck = torch.load(path, map_location='cpu', weights_only=True)
    assert {'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'scaler_state_dict', 'rng_state', 'ema_state_dict'} <= set(ck)
    assert any(k.startswith('model.tf_layers.0.') for k in ck['model_state_dict']) and 'initted' in ck['ema_state_dict']
    _, model2, _ = ema_model()
    model2.train()
    opt2 = FlatAdamW(model2.model.parameters(), lr=2e-3)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=2, gamma=0.5)
    CK.load_all_state_dicts(model2.cfg, model2, opt2, sched2, None, path)
    assert opt2.steps == 3 and opt2.lr == opt.lr and model2.ema._host_step == 3
    # the same random draws (t is fixed; the noise of apply_noise comes from torch's generator) on both continuations
    torch.manual_seed(2); run(model, opt, sched, 3)
    torch.manual_seed(2); run(model2, opt2, sched2, 3)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=0, atol=5e-6), k
    assert torch.allclose(opt.exp_avg, opt2.exp_avg, atol=1e-6) and torch.allclose(opt.max_exp_avg_sq, opt2.max_exp_avg_sq, atol=1e-6)


def test_train_one_epoch_over_size_binned_batches():
    """training/helpers.py:121-152 on three batches of different (B, n) (the reference bins graphs by size), two epochs."""
    from diffalign_amd.optim import FlatAdamW
    from diffalign_amd.training import train_one_epoch
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    z, model, _ = ema_model()
    model.pos_encodings_fn = None
    model.cfg.neuralnet.pos_encoding_type = 'smiles_pos_enc'

    def batch(B, n, seed):
        d = S.make_reaction_batch(B, n, seed=seed)
        return PlaceHolder(X=d['X'].to(D), E=d['E'].to(D), y=torch.zeros(B, 1, device=D), node_mask=d['node_mask'].to(D),
                           atom_map_numbers=d['atom_map_numbers'].to(D), mol_assignments=d['mol_assignments'].to(D))
    batches = [batch(4, 24, 1), batch(3, 40, 2), batch(2, 33, 3)]
    opt = FlatAdamW(model.model.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    losses = []
    torch.manual_seed(0)
    for epoch in range(4):
        train_one_epoch(model, batches, opt, sched, epoch=epoch, losses=losses)
    assert len(losses) == 4 and all(l == l and l < 1e4 for l in losses) and losses[-1] < losses[0], losses
    assert abs(opt.lr - 2e-3 * 0.5 ** 4) < 1e-12 and opt.steps == 12 and int(model.ema.step) == 12
