"""Transition tables with the reference's class names and constructor arguments
(diffalign/diffusion/noise_schedule.py:43-326), built from the formulas (not copied code) and kept on an
explicit device (the reference uses a module-level `device` global, noise_schedule.py:10,134)."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .placeholder import PlaceHolder


def cosine_beta_schedule_discrete(timesteps: int, s: float = 0.008) -> np.ndarray:
    """utils/diffusion/helpers.py:634-643."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    return (1 - ac[1:] / ac[:-1]).squeeze()


def _cosine_cumprod(timesteps: int, s: float) -> np.ndarray:
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    return ac / ac[0]


def cosine_beta_schedule_discrete_alternate(timesteps: int) -> np.ndarray:
    """utils/diffusion/helpers.py:645-656: betas = 1 - the cumulative product itself (zero transition probability at t = 0; the
    reference notes that the overall shape differs from the cosine schedule)."""
    return (1 - _cosine_cumprod(timesteps, 0.0)).squeeze()


def cosine_beta_schedule_discrete_alternate_2(timesteps: int) -> np.ndarray:
    """utils/diffusion/helpers.py:658-670: the s = 0 cosine ratios behind a leading alpha of 1."""
    ac = _cosine_cumprod(timesteps, 0.0)
    return (1 - np.concatenate([np.array([1.0]), ac[1:] / ac[:-1]])).squeeze()


def custom_beta_schedule_discrete(timesteps: int, average_num_nodes: int = 50, s: float = 0.008) -> np.ndarray:
    """utils/diffusion/helpers.py:672-692: cosine betas floored at 1.2 expected edge updates per graph and step."""
    assert timesteps >= 100
    betas = cosine_beta_schedule_discrete(timesteps, s).copy()
    num_edges = average_num_nodes * (average_num_nodes - 1) / 2
    beta_first = 1.2 / ((4 / 5) * num_edges)
    betas[betas < beta_first] = beta_first
    return np.array(betas)


class PredefinedNoiseScheduleDiscrete(torch.nn.Module):
    """noise_schedule.py:43-86."""

    def __init__(self, noise_schedule: str, timesteps: int):
        super().__init__()
        self.timesteps = timesteps
        if noise_schedule == 'cosine':
            b = cosine_beta_schedule_discrete(timesteps, s=0)
        elif noise_schedule == 'cosine_alternate':
            b = cosine_beta_schedule_discrete_alternate(timesteps)
        elif noise_schedule == 'cosine_alternate_2':
            b = cosine_beta_schedule_discrete_alternate_2(timesteps)
        elif noise_schedule == 'custom':
            b = custom_beta_schedule_discrete(timesteps)
        else:
            raise NotImplementedError(noise_schedule)
        betas = torch.from_numpy(b).float()
        self.register_buffer('betas', betas)
        self.alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(self.alphas), dim=0))

    def forward(self, t_normalized=None, t_int=None):
        assert int(t_normalized is None) + int(t_int is None) == 1
        if t_int is None:
            t_int = torch.round(t_normalized * self.timesteps)
        return self.betas[t_int.long()]

    def get_alpha_bar(self, t_normalized=None, t_int=None):
        assert int(t_normalized is None) + int(t_int is None) == 1
        if t_int is None:
            t_int = torch.round(t_normalized * self.timesteps)
        return self.alphas_bar[t_int.long()]


def _as_list(idx) -> Optional[list]:
    if idx is None:
        return None
    return [int(i) for i in (idx.tolist() if isinstance(idx, torch.Tensor) else idx)]


class DiffusionTransition:
    """noise_schedule.py:88-169.  Subclasses fill Qts_*/Qt_bars_*/*_limit before calling this ctor."""

    def __init__(self, x_classes, e_classes, y_classes, diffuse_edges=True, node_idx_to_mask=(1,), edge_idx_to_mask=None,
                 device='cpu'):
        self.x_classes, self.e_classes, self.y_classes = x_classes, e_classes, y_classes
        self.diffuse_edges = diffuse_edges
        nm, em = _as_list(node_idx_to_mask), _as_list(edge_idx_to_mask)

        def fix(Q, idx):
            if idx is not None:
                Q[..., :, idx] = 0.
                Q[..., idx, :] = 0.
                Q[..., idx, idx] = 1.
            return Q / Q.sum(dim=-1, keepdim=True)

        self.Qts_x, self.Qt_bars_x = fix(self.Qts_x, nm), fix(self.Qt_bars_x, nm)
        self.Qts_e, self.Qt_bars_e = fix(self.Qts_e, em), fix(self.Qt_bars_e, em)
        if nm is not None:
            self.x_limit[nm] = 0.
        if em is not None:
            self.e_limit[em] = 0.
        self.x_limit = self.x_limit / self.x_limit.sum()
        self.e_limit = self.e_limit / self.e_limit.sum()
        self.to_device(device)

    def to_device(self, device):
        for name in ('Qts_x', 'Qts_e', 'Qts_y', 'Qt_bars_x', 'Qt_bars_e', 'Qt_bars_y', 'x_limit', 'e_limit', 'y_limit'):
            setattr(self, name, getattr(self, name).to(device).contiguous())
        return self

    def _idx(self, t):
        return t[:, 0].long().to(self.Qts_x.device)

    def get_Qt(self, t, device=None, diffuse_edges=True):
        idx = self._idx(t)
        E = self.Qts_e[idx] if self.diffuse_edges else \
            torch.eye(self.e_classes, device=self.Qts_x.device)[None].repeat(t.shape[0], 1, 1)
        return PlaceHolder(X=self.Qts_x[idx], E=E, y=self.Qts_y[idx])

    def get_Qt_bar(self, t, device=None, diffuse_edges=True):
        idx = self._idx(t)
        E = self.Qt_bars_e[idx] if self.diffuse_edges else \
            torch.eye(self.e_classes, device=self.Qts_x.device)[None].repeat(t.shape[0], 1, 1)
        return PlaceHolder(X=self.Qt_bars_x[idx], E=E, y=self.Qt_bars_y[idx])

    def get_limit_dist(self):
        return PlaceHolder(X=self.x_limit, E=self.e_limit, y=self.y_limit)

    # -- one reverse step whose B graphs share the step number (the sampling loop): the three per-graph table sets a step needs.
    # get_Qt / get_Qt_bar cost a dozen gather kernels (6 us each, serial, in front of every step); here every step's (B,K,K) tables
    # are laid out once per batch size, (T+1, B, K, K) per family, and a step takes slices: no kernel, no stream, no event.
    # (Preparing only the next step's set on a side stream was tried first: handing its tensors to the caller's stream costs one
    # allocator event record per tensor on that stream, ~6 us each -- as much as the gathers it saved.)
    STEP_TABLE_BYTES_MAX = 512 << 20
    STEP_STATS = {'sliced': 0, 'built': 0}                   # how step_tables was served

    def _step_tables_now(self, t: int, B: int):
        rep = lambda Q, i: Q[i].expand(B, -1, -1).contiguous()
        eye = (lambda: torch.eye(self.e_classes, device=self.Qts_x.device)[None].repeat(B, 1, 1))
        E = (lambda Q, i: rep(Q, i)) if self.diffuse_edges else (lambda Q, i: eye())
        return (PlaceHolder(X=rep(self.Qts_x, t), E=E(self.Qts_e, t), y=None),
                PlaceHolder(X=rep(self.Qt_bars_x, t - 1), E=E(self.Qt_bars_e, t - 1), y=None),
                PlaceHolder(X=rep(self.Qt_bars_x, t), E=E(self.Qt_bars_e, t), y=None))

    def step_tables(self, t: int, B: int):
        """(Q_t, Qbar_{t-1}, Qbar_t) as (B,K,K) tables for step t shared by all B graphs (== get_Qt / get_Qt_bar on a constant t
        array, without the y tables)."""
        T1 = self.Qts_x.shape[0]
        assert 1 <= t < T1
        nbytes = 2 * T1 * B * 4 * (self.x_classes ** 2 + self.e_classes ** 2)
        if nbytes > self.STEP_TABLE_BYTES_MAX or not self.diffuse_edges:
            DiffusionTransition.STEP_STATS['built'] += 1
            return self._step_tables_now(t, B)
        cache = self.__dict__.setdefault('_step_all', {})
        if B not in cache:
            if self.Qts_x.is_cuda and torch.cuda.is_current_stream_capturing():
                return self._step_tables_now(t, B)
            cache.clear()                                            # one batch size at a time
            rep = lambda Q: Q[:, None].expand(-1, B, -1, -1).contiguous()
            cache[B] = (rep(self.Qts_x), rep(self.Qts_e), rep(self.Qt_bars_x), rep(self.Qt_bars_e))
        qx, qe, bx, be = cache[B]
        DiffusionTransition.STEP_STATS['sliced'] += 1
        return (PlaceHolder(X=qx[t], E=qe[t], y=None), PlaceHolder(X=bx[t - 1], E=be[t - 1], y=None), PlaceHolder(X=bx[t], E=be[t], y=None))


def _mix_family(T, u, betas, abar):
    K = u.shape[-1]
    eye = torch.eye(K)
    Qt = torch.stack([betas[t] * u + (1 - betas[t]) * eye for t in range(T + 1)])
    Qb = torch.stack([abar[t] * eye + (1 - abar[t]) * u for t in range(T + 1)])
    return Qt, Qb


class MarginalUniformTransition(DiffusionTransition):
    """noise_schedule.py:234-292: Q_t=(1-b_t)I+b_t 1 m^T, Qbar_t=abar_t I+(1-abar_t) 1 m^T."""

    def __init__(self, x_marginals, e_marginals, y_classes, noise_schedule, timesteps, diffuse_edges=True,
                 node_idx_to_mask=None, edge_idx_to_mask=None, device='cpu'):
        self.predefined_noise_schedule = PredefinedNoiseScheduleDiscrete(noise_schedule, timesteps)
        self.noise_schedule, self.timesteps = noise_schedule, timesteps
        self.x_marginals, self.e_marginals = x_marginals.float().cpu(), e_marginals.float().cpu()
        b, ab = self.predefined_noise_schedule.betas, self.predefined_noise_schedule.alphas_bar
        kx, ke = len(x_marginals), len(e_marginals)
        self.Qts_x, self.Qt_bars_x = _mix_family(timesteps, self.x_marginals[None, :].expand(kx, -1), b, ab)
        self.Qts_e, self.Qt_bars_e = _mix_family(timesteps, self.e_marginals[None, :].expand(ke, -1), b, ab)
        self.Qts_y, self.Qt_bars_y = _mix_family(timesteps, torch.ones(y_classes, y_classes) / max(y_classes, 1), b, ab)
        self.x_limit, self.e_limit = self.x_marginals.clone(), self.e_marginals.clone()
        self.y_limit = torch.ones(y_classes) / max(y_classes, 1)
        super().__init__(kx, ke, y_classes, diffuse_edges, node_idx_to_mask, edge_idx_to_mask, device)


class DiscreteUniformTransition(MarginalUniformTransition):
    """noise_schedule.py:171-232."""

    def __init__(self, noise_schedule, timesteps, x_classes, e_classes, y_classes, diffuse_edges=True,
                 node_idx_to_mask=(3,), edge_idx_to_mask=None, device='cpu'):
        super().__init__(torch.ones(x_classes) / x_classes, torch.ones(e_classes) / e_classes, y_classes, noise_schedule,
                         timesteps, diffuse_edges, node_idx_to_mask, edge_idx_to_mask, device)


class AbsorbingStateTransitionMaskNoEdge(DiffusionTransition):
    """noise_schedule.py:294-326: Q_t=(1-b)I+b 1 e_abs^T with b=1/(T-t+1); Qbar_t=(1-t/T)I+(t/T) 1 e_abs^T."""

    def __init__(self, timesteps, x_classes, e_classes, y_classes, diffuse_edges=True, abs_state_position_x=-1,
                 abs_state_position_e=0, node_idx_to_mask=(3,), edge_idx_to_mask=None, device='cpu'):
        T = timesteps
        self.timesteps = T

        def fam(K, a):
            eye = torch.eye(K)
            absorb = torch.zeros(K, K); absorb[:, a] = 1.
            Qt = [eye] + [1 / (T - t + 1) * absorb + (1 - 1 / (T - t + 1)) * eye for t in range(1, T + 1)]
            Qb = [eye] + [t / T * absorb + (1 - t / T) * eye for t in range(1, T + 1)]
            lim = torch.zeros(K); lim[a] = 1.
            return torch.stack(Qt), torch.stack(Qb), lim

        self.Qts_x, self.Qt_bars_x, self.x_limit = fam(x_classes, abs_state_position_x % x_classes)
        self.Qts_e, self.Qt_bars_e, self.e_limit = fam(e_classes, abs_state_position_e % e_classes)
        self.Qts_y, self.Qt_bars_y, self.y_limit = fam(y_classes + 1, y_classes)
        super().__init__(x_classes, e_classes, y_classes, diffuse_edges, node_idx_to_mask, edge_idx_to_mask, device)
