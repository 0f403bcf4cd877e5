"""diffalign_amd.ema.EMA against the reference's EMA class (neuralnet/ema_pytorch.py) run on the same weight trajectory
(tests/golden/ema.npz, tests/golden/make_golden_r2.py gen_ema): same decay schedule, same EMA weights (1e-6), same state-dict
keys, and a checkpoint round trip.  CPU (the update is device-agnostic torch multi-tensor ops); the GPU wiring is in
tests/test_gpu_ema.py."""
import numpy as np
import torch
import torch.nn as nn

from tests.golden.common import load


def _net(z):
    net = nn.Sequential(nn.Linear(5, 4), nn.ReLU(), nn.Linear(4, 3))
    net.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('init::')})
    return net


def _run(ema, net, z, lo, hi, decays=None):
    for t in range(lo, hi):
        with torch.no_grad():
            for k, p in net.named_parameters():
                p.add_(torch.from_numpy(z['inc::' + k][t]))
        ema.update()
        if decays is not None:
            decays.append(ema.get_current_decay())


def test_ema_matches_reference_trajectory():
    from diffalign_amd.ema import EMA
    z = load('ema')
    net = _net(z)
    ema = EMA(net, beta=0.99, power=1)
    assert sorted(ema.state_dict().keys()) == [str(k) for k in z['state_keys']]
    decays, last = [], 0
    for cp in (100, 111, 121, 160):
        _run(ema, net, z, last, cp, decays)
        last = cp
        for k, v in ema.ema_model.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f'ema{cp}::' + k], rtol=0, atol=1e-6, err_msg=f'{cp} {k}')
    np.testing.assert_allclose(np.array(decays, dtype=np.float32), z['decays'], rtol=1e-6)
    assert int(ema.step) == int(z['final_step'][0]) and float(ema.initted) == float(z['final_initted'][0])
    assert not any(p.requires_grad for p in ema.ema_model.parameters())


def test_ema_checkpoint_round_trip_and_exclusions():
    from diffalign_amd.ema import EMA
    z = load('ema')
    net = _net(z)
    ema = EMA(net, beta=0.99, power=1)
    _run(ema, net, z, 0, 130)
    sd = {k: v.clone() for k, v in ema.state_dict().items()}
    net2 = _net(z)
    ema2 = EMA(net2, beta=0.99, power=1)
    ema2.load_state_dict(sd)                                   # also restores the online model and the host-side counters
    _run(ema, net, z, 130, 160); _run(ema2, net2, z, 130, 160)
    for (k, a), (_, b) in zip(ema.ema_model.state_dict().items(), ema2.ema_model.state_dict().items()):
        assert torch.equal(a, b), k
        np.testing.assert_allclose(a.numpy(), z['ema160::' + k], rtol=0, atol=1e-6)
    # names excluded from averaging are copied, ignored names stay as they were
    net3 = _net(z)
    ema3 = EMA(net3, beta=0.99, power=1, update_after_step=0, update_every=1, param_or_buffer_names_no_ema={'0.bias'}, ignore_names={'2.bias'})
    start = ema3.ema_model.state_dict()['2.bias'].clone()
    _run(ema3, net3, z, 0, 20)
    assert torch.equal(ema3.ema_model.state_dict()['0.bias'], net3.state_dict()['0.bias'])
    first = start + torch.from_numpy(z['inc::2.bias'][0]) + torch.from_numpy(z['inc::2.bias'][1])   # copied at steps 0 and 1 (first averaged update), never touched again
    assert torch.allclose(ema3.ema_model.state_dict()['2.bias'], first)
    assert not torch.equal(ema3.ema_model.state_dict()['0.weight'], net3.state_dict()['0.weight'])
