"""Host-side transition tables (diffalign_amd/noise_schedule.py) against the reference's classes for every schedule name the
reference accepts (diffusion/noise_schedule.py:52-61: cosine, cosine_alternate, cosine_alternate_2, custom) and the three transition
families (tests/golden/schedules.npz, transitions.npz).  CPU: the tables are plain torch on an explicit device."""
import numpy as np
import pytest
import torch

from tests.golden.common import load, ATOMS, BONDS

KX, KE = len(ATOMS), len(BONDS)
T = lambda a: torch.from_numpy(np.asarray(a))
NM, EM = [28, 0], [4, 5, 6]


@pytest.mark.parametrize('name', ['cosine', 'cosine_alternate', 'cosine_alternate_2', 'custom'])
def test_schedules_and_tables_match_reference(name):
    from diffalign_amd import noise_schedule as NS
    z = load('schedules')
    sch = NS.PredefinedNoiseScheduleDiscrete(name, 100)
    assert (sch.betas - T(z[f'{name}::betas'])).abs().max() <= 1e-7
    assert (sch.alphas_bar - T(z[f'{name}::alphas_bar'])).abs().max() <= 1e-6
    t = torch.tensor([[3], [50]])
    assert torch.equal(sch(t_int=t), sch.betas[t]) and torch.equal(sch.get_alpha_bar(t_normalized=t / 100), sch.alphas_bar[t])
    sel = z['sel'].tolist()
    u = NS.DiscreteUniformTransition(name, 100, KX, KE, 1, node_idx_to_mask=NM, edge_idx_to_mask=EM, device='cpu')
    m = NS.MarginalUniformTransition(T(z['x_marginals']), T(z['e_marginals']), 1, name, 100, node_idx_to_mask=NM, edge_idx_to_mask=EM,
                                     device='cpu')
    for tag, tm in (('uni', u), ('mar', m)):
        for f in ('Qts_x', 'Qts_e', 'Qt_bars_x', 'Qt_bars_e'):
            got, want = getattr(tm, f)[sel], T(z[f'{name}::{tag}_{f}'])
            assert (got - want).abs().max() <= 2e-6, (tag, f, float((got - want).abs().max()))
            assert (got >= 0).all() and (got.sum(-1) - 1).abs().max() <= 1e-5           # rows are distributions
        assert (tm.x_limit - T(z[f'{name}::{tag}_x_limit'])).abs().max() <= 1e-6
        assert (tm.e_limit - T(z[f'{name}::{tag}_e_limit'])).abs().max() <= 1e-6


def test_unknown_schedule_is_refused_and_ten_step_tables():
    from diffalign_amd import noise_schedule as NS
    with pytest.raises(NotImplementedError):
        NS.PredefinedNoiseScheduleDiscrete('linear', 100)
    z = load('transitions')
    u = NS.DiscreteUniformTransition('cosine', 10, KX, KE, 1, node_idx_to_mask=NM, edge_idx_to_mask=EM, device='cpu')
    a = NS.AbsorbingStateTransitionMaskNoEdge(10, KX, KE, 1, abs_state_position_x=27, abs_state_position_e=0, node_idx_to_mask=NM,
                                              edge_idx_to_mask=EM, device='cpu')
    for tag, tm in (('uni10', u), ('abs10', a)):
        for f in ('Qts_x', 'Qts_e', 'Qt_bars_x', 'Qt_bars_e', 'x_limit', 'e_limit'):
            assert (getattr(tm, f) - T(z[f'{tag}_{f}'])).abs().max() <= 1e-6, (tag, f)
    t = T(z['abs10_gather_t'])
    assert (a.get_Qt(t).X - T(z['abs10_gather_Qt_x'])).abs().max() <= 1e-6
    assert (a.get_Qt_bar(t).E - T(z['abs10_gather_Qtb_e'])).abs().max() <= 1e-6
