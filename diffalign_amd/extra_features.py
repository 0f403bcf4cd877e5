"""Extra input features of the denoiser on the device.

Host-side mirror of the reference's `ExtraFeatures` (neuralnet/extra_features.py:22-72) and `ExtraMolecularFeatures`
(neuralnet/extra_features_molecular.py:6-26): same constructor arguments, same `forward` signatures and return
tuples.  The reference moves E, X and node_mask to the CPU at every step (diffusion_abstract.py:361-364); here everything
stays on the device: HIP kernels (csrc/k_extra.hip) for the passes over the edge state and the per-graph reductions,
`torch.bmm` for the two adjacency powers and, for the Laplacian, a one-launch batched partial eigensolver in fp64
(csrc/k_extra_eig.hip; `DA_EXTRA_EIGH=torch` selects `torch.linalg.eigh`, the reference's routine, instead).

Requires a symmetric E (every diffusion state is: helpers.py:743-745) -- the cycle formulas use A = A^T.
The eigenvector features (not-in-largest-component flag, two lowest non-trivial eigenvectors) depend on the basis the
eigensolver returns inside degenerate eigenspaces, in the reference as here; eigenvalue features do not."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L


def _i(v):
    return C.c_int(int(v))


def _adjacency(E: torch.Tensor, nm_u8: torch.Tensor, bond_orders: torch.Tensor, want_L: bool):
    B, n, _, ke = E.shape
    dev = E.device
    A = torch.empty(B, n, n, device=dev)
    Lap = torch.empty(B, n, n, dtype=torch.float64, device=dev) if want_L else None
    vw = torch.empty(B, n, device=dev); vp = torch.empty(B, n, device=dev)
    L.check(L.lib().da_extra_adjacency(L.ptr(E, torch.float32), L.ptr(nm_u8), L.ptr(bond_orders, torch.float32), L.ptr(A), L.ptr(Lap),
                                       L.ptr(vw), L.ptr(vp), _i(B), _i(n), _i(ke), L.stream()), 'da_extra_adjacency')
    return A, Lap, vw, vp


def _cycles(A: torch.Tensor, nm_u8: torch.Tensor):
    B, n, _ = A.shape
    k2 = torch.bmm(A, A)
    k3 = torch.bmm(k2, A)
    x = torch.empty(B, n, 3, device=A.device); y = torch.empty(B, 4, device=A.device)
    L.check(L.lib().da_extra_cycles(L.ptr(A), L.ptr(k2), L.ptr(k3), L.ptr(nm_u8), L.ptr(x), L.ptr(y), _i(B), _i(n), L.stream()),
            'da_extra_cycles')
    return x, y


# 'hip' (default): the one-launch batched partial eigensolver of csrc/k_extra_eig.hip; 'torch': torch.linalg.eigh (rocSOLVER
# syevd, fp64, the reference's routine) followed by k_extra_eigen -- 3.5x slower at B = 64, n = 300, kept for cross-checks.
EIGENSOLVER = os.environ.get('DA_EXTRA_EIGH', 'hip')
EIG_CACHE = os.environ.get('DA_EIG_CACHE', '1') != '0'        # A/B switch: the fixed block's reflectors kept along a chain (EigCache)


class EigCache:
    """What the eigensolver may keep between calls on the SAME sequence of graphs (a reverse chain): the nodes from
    `fixed_from[b]` on keep all their edges at every step (graph.apply_mask: every pair with a fixed endpoint stays at the
    condition's value) -- the product side of a reaction, roughly half of the matrix and, with absorbing noise, most of its
    connected part.  The first call tridiagonalises everything and keeps that block's reflectors; later calls reduce only the
    rows before it (da_extra_eigen_solve_cached: same bits as the uncached solver).  One object per sample_one_batch call."""

    def __init__(self, fixed_from: torch.Tensor, n: int):
        B = fixed_from.shape[0]
        dev = fixed_from.device
        self.n = n
        self.fixed_from = fixed_from.to(torch.int32).clamp(0, n).contiguous()
        self.mmax = int(n - int(self.fixed_from.min())) if B else 0          # one host sync per chain, before its first step
        self.state = torch.zeros(B, dtype=torch.int32, device=dev)
        self.store = torch.empty(B, self.mmax + 2, self.mmax, dtype=torch.float64, device=dev) if self.mmax > 0 else None

    @staticmethod
    def from_free_masks(free_e: torch.Tensor, node_mask: torch.Tensor) -> 'EigCache':
        """free_e (B,n): 1 where the node's edges may change (graph.get_node_masks); padding never changes.  The fixed block starts
        behind the last node that is real and free."""
        B, n = node_mask.shape
        moving = (free_e != 0) & (node_mask != 0)
        idx = torch.arange(1, n + 1, device=node_mask.device)[None, :].expand(B, n)
        return EigCache(torch.where(moving, idx, torch.zeros_like(idx)).amax(dim=1), n)


_EIG_STREAMS = {}
EIG_SIDE_STREAM = os.environ.get('DA_EIG_SIDE_STREAM', '1') != '0'      # A/B switch


def _eig_stream(dev):
    """Stream of the eigensolver beside the other feature kernels (none under HIP-graph capture)."""
    if not EIG_SIDE_STREAM or torch.cuda.is_current_stream_capturing():
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _EIG_STREAMS:
        _EIG_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _EIG_STREAMS[key]


def _eigen(Lap: torch.Tensor, nm_u8: torch.Tensor, pad_noise: Optional[torch.Tensor], solver: Optional[str] = None,
           cache: Optional[EigCache] = None):
    """Eigen-features of the fp64 Laplacians (B,n,n) (extra_features.py:117-128, :158-204).  `Lap` is overwritten."""
    B, n, _ = Lap.shape
    dev = Lap.device
    if pad_noise is None:
        pad_noise = torch.randn(B, n, device=dev)             # :189 (only keeps 0 from becoming the mode on padded slots)
    pad_noise = pad_noise.contiguous()
    ncomp = torch.empty(B, device=dev); ev5 = torch.empty(B, 5, device=dev)
    nonlcc = torch.empty(B, n, device=dev); vec2 = torch.empty(B, n, 2, device=dev)
    if (solver or EIGENSOLVER) == 'hip' and n <= 1024:
        if cache is not None and cache.store is not None and EIG_CACHE:
            assert cache.n == n and cache.state.shape[0] == B and cache.state.device == dev, 'EigCache belongs to another batch'
            L.check(L.lib().da_extra_eigen_solve_cached(L.ptr(Lap, torch.float64), L.ptr(nm_u8), L.ptr(pad_noise, torch.float32), L.ptr(ncomp),
                                                        L.ptr(ev5), L.ptr(nonlcc), L.ptr(vec2), C.c_void_p(0), _i(B), _i(n),
                                                        L.ptr(cache.fixed_from, torch.int32), L.ptr(cache.state, torch.int32),
                                                        L.ptr(cache.store, torch.float64), _i(cache.mmax), L.stream()),
                    'da_extra_eigen_solve_cached')
            return ncomp, ev5, nonlcc, vec2
        L.check(L.lib().da_extra_eigen_solve(L.ptr(Lap, torch.float64), L.ptr(nm_u8), L.ptr(pad_noise, torch.float32), L.ptr(ncomp), L.ptr(ev5),
                                             L.ptr(nonlcc), L.ptr(vec2), C.c_void_p(0), _i(B), _i(n), L.stream()), 'da_extra_eigen_solve')
        return ncomp, ev5, nonlcc, vec2
    evals, evecs = torch.linalg.eigh(Lap)                     # extra_features.py:118 (fp64)
    evals = evals.contiguous()
    s = evecs.stride()
    assert s[0] == n * n, 'eigenvector batch is not dense'
    if not evecs.is_cuda:
        raise L.DiffAlignHipError('diffalign_amd ops need tensors on a HIP device (no CPU fallback)')
    L.check(L.lib().da_extra_eigen(L.ptr(evals, torch.float64), C.c_void_p(evecs.data_ptr()), C.c_longlong(s[1]), C.c_longlong(s[2]),
                                   L.ptr(nm_u8), L.ptr(pad_noise, torch.float32), L.ptr(ncomp), L.ptr(ev5), L.ptr(nonlcc),
                                   L.ptr(vec2), _i(B), _i(n), L.stream()), 'da_extra_eigen')
    return ncomp, ev5, nonlcc, vec2


class ExtraFeatures(nn.Module):
    """neuralnet/extra_features.py:22-72.  forward(E, node_mask) -> (X_res, E_res, y_res)."""

    def __init__(self, extra_features_type, dataset_info):
        super().__init__()
        if extra_features_type not in ('cycles', 'eigenvalues', 'all'):
            raise ValueError(f'Features type {extra_features_type} not implemented')
        self.max_n_nodes = dataset_info.max_n_nodes
        self.features_type = extra_features_type

    def forward(self, E, node_mask, pad_noise: Optional[torch.Tensor] = None,
                eig_cache: Optional[EigCache] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        B, n = node_mask.shape
        dev = E.device
        E = E.contiguous().float()
        nm_u8 = L.as_u8(node_mask)
        zeros = torch.zeros(E.shape[-1], device=dev)          # the valency sums are not needed here
        A, Lap, _, _ = _adjacency(E, nm_u8, zeros, self.features_type != 'cycles')
        x_cycles, y_cycles = _cycles(A, nm_u8)
        n_frac = node_mask.sum(dim=1, keepdim=True).float() / self.max_n_nodes
        E_res = torch.zeros((*E.shape[:-1], 0), device=dev, dtype=E.dtype)
        if self.features_type == 'cycles':
            return x_cycles, E_res, torch.hstack((n_frac, y_cycles))
        ncomp, ev5, nonlcc, vec2 = _eigen(Lap, nm_u8, pad_noise, cache=eig_cache)
        y_res = torch.hstack((n_frac, y_cycles, ncomp.unsqueeze(-1), ev5))
        if self.features_type == 'eigenvalues':
            return x_cycles, E_res, y_res
        return torch.cat((x_cycles, nonlcc.unsqueeze(-1), vec2), dim=-1), E_res, y_res


class ExtraMolecularFeatures(nn.Module):
    """neuralnet/extra_features_molecular.py:6-26.  forward(X, E) -> (X_res (B,n,2): charge, valency; E_res; y_res (B,1): weight)."""

    def __init__(self, dataset_infos):
        super().__init__()
        self.valencies = [float(v) for v in dataset_infos.valencies]
        self.bond_orders = [float(v) for v in dataset_infos.bond_orders]
        self.atom_weight_list = [float(v) for v in dataset_infos.atom_weights.values()]
        self.max_weight = float(dataset_infos.max_weight)
        self._consts = {}

    def consts(self, dev):
        if dev not in self._consts:
            self._consts[dev] = tuple(torch.tensor(v, dtype=torch.float32, device=dev)
                                      for v in (self.valencies, self.bond_orders, self.atom_weight_list))
        return self._consts[dev]

    def from_valency_sums(self, X, vw, vp):
        B, n, kx = X.shape
        val, _, aw = self.consts(X.device)
        assert val.numel() == kx and aw.numel() == kx, 'dataset_info.valencies / atom_weights do not match the node classes'
        cv = torch.empty(B, n, 2, device=X.device); w = torch.empty(B, device=X.device)
        L.check(L.lib().da_extra_molecular(L.ptr(X.contiguous().float(), torch.float32), L.ptr(val), L.ptr(aw), L.ptr(vw), L.ptr(vp),
                                           C.c_float(self.max_weight), L.ptr(cv), L.ptr(w), _i(B), _i(n), _i(kx), L.stream()),
                'da_extra_molecular')
        return cv, w.unsqueeze(-1)

    def forward(self, X, E):
        B, n = X.shape[:2]
        E = E.contiguous().float()
        nm_u8 = torch.ones(B, n, dtype=torch.uint8, device=X.device)
        _, _, vw, vp = _adjacency(E, nm_u8, self.consts(X.device)[1], False)
        cv, w = self.from_valency_sums(X, vw, vp)
        return cv, torch.zeros((*E.shape[:-1], 0), device=E.device, dtype=E.dtype), w


def _eigen_into(Lap, nm_u8, pad_noise, Xf, x_col, yf, y_col, cache: Optional[EigCache]):
    """_eigen with the six features written by the solver itself into columns x_col.. of Xf's rows (nonlcc, vec2) and y_col.. of
    yf's rows (n_comp, ev5): nothing is left to do behind the solver (da_extra_eigen_solve_into).  The library path
    (`DA_EXTRA_EIGH=torch`, n > 1024) copies."""
    B, n, _ = Lap.shape
    if EIGENSOLVER != 'hip' or n > 1024:
        ncomp, ev5, nonlcc, vec2 = _eigen(Lap, nm_u8, pad_noise, cache=cache)
        Xf[..., x_col] = nonlcc; Xf[..., x_col + 1:x_col + 3] = vec2
        yf[:, y_col] = ncomp; yf[:, y_col + 1:y_col + 6] = ev5
        return
    assert Xf.is_cuda and Xf.is_contiguous() and yf.is_contiguous() and Xf.dtype == torch.float32 and yf.dtype == torch.float32
    use = cache is not None and cache.store is not None and EIG_CACHE
    if use:
        assert cache.n == n and cache.state.shape[0] == B and cache.state.device == Lap.device, 'EigCache belongs to another batch'
    null = C.c_void_p(0)
    L.check(L.lib().da_extra_eigen_solve_into(L.ptr(Lap, torch.float64), L.ptr(nm_u8), L.ptr(pad_noise.contiguous(), torch.float32),
                                              C.c_void_p(Xf.data_ptr() + 4 * x_col), _i(Xf.shape[-1]),
                                              C.c_void_p(yf.data_ptr() + 4 * y_col), _i(yf.shape[-1]), _i(B), _i(n),
                                              L.ptr(cache.fixed_from, torch.int32) if use else null,
                                              L.ptr(cache.state, torch.int32) if use else null,
                                              L.ptr(cache.store, torch.float64) if use else null, _i(cache.mmax if use else 0), L.stream()),
            'da_extra_eigen_solve_into')


def compute_extra_features(X, E, node_mask, extra: ExtraFeatures, molecular: ExtraMolecularFeatures,
                           pad_noise: Optional[torch.Tensor] = None, eig_cache: Optional[EigCache] = None,
                           base: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, defer_join: bool = False):
    """The two calculators of diffusion_abstract.py:350-372 with ONE pass over the edge state.
    Returns (X_extra (B,n,8), y_extra (B,12)) in the reference's concatenation order (features_type 'all'), or, with
    base = (X0, y0), the denoiser's widened inputs (cat(X0, X_extra), cat(y0, y_extra)) of :366-371.  The rows are assembled in
    place: everything but the six eigen-features sits in its columns while the eigensolver still runs, and the solver writes
    those itself -- behind it, on the step's critical path, nothing is left (seven small torch launches before).
    `defer_join`: return (Xf, yf, pending) without making the caller's stream wait for the eigensolver's stream; pending = (event,
    tensors the solver still uses) or None.  Whoever reads Xf / yf waits for the event on its stream, and the CALLER'S stream must wait
    for it too before `pending` is dropped (the tensors were allocated on it): engine.model_forward(step['inputs_ready']) does both,
    with the edge input stage -- which reads none of this -- enqueued in front of the wait."""
    assert extra.features_type == 'all'
    dev = E.device
    B, n = node_mask.shape
    E = E.contiguous().float()
    nm_u8 = L.as_u8(node_mask)
    X0, y0 = base if base is not None else (None, None)
    kx0, ky0 = (X0.shape[-1], y0.shape[-1]) if base is not None else (0, 0)
    Xf = torch.empty(B, n, kx0 + 8, device=dev); yf = torch.empty(B, ky0 + 12, device=dev)
    A, Lap, vw, vp = _adjacency(E, nm_u8, molecular.consts(dev)[1], True)
    if pad_noise is None:
        pad_noise = torch.randn(node_mask.shape, device=dev)

    def rest():                     # cycle counts (two batched GEMMs and a kernel), molecular features, the copies of X0 / y0
        x_cycles, y_cycles = _cycles(A, nm_u8)
        cv, w = molecular.from_valency_sums(X, vw, vp)
        if base is not None:
            Xf[..., :kx0] = X0; yf[:, :ky0] = y0
        Xf[..., kx0:kx0 + 3] = x_cycles; Xf[..., kx0 + 6:] = cv
        yf[:, ky0] = node_mask.sum(dim=1).float() / extra.max_n_nodes
        yf[:, ky0 + 1:ky0 + 5] = y_cycles; yf[:, ky0 + 11:] = w

    # the eigensolver runs one workgroup per graph (64 of 256 CUs at the benchmark batch): everything else runs beside it on the
    # caller's stream
    side = _eig_stream(dev)
    if side is None:
        _eigen_into(Lap, nm_u8, pad_noise, Xf, kx0 + 3, yf, ky0 + 5, eig_cache)
        rest()
    else:
        main = torch.cuda.current_stream()
        ready = main.record_event()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            _eigen_into(Lap, nm_u8, pad_noise, Xf, kx0 + 3, yf, ky0 + 5, eig_cache)
            done = side.record_event()
        rest()
        if defer_join:
            return Xf, yf, (done, (Lap, A, nm_u8, pad_noise, Xf, yf))
        main.wait_event(done)       # (Xf, yf, Lap, pad_noise: allocated on the caller's stream, which has now waited for their last use)
    return (Xf, yf, None) if defer_join else (Xf, yf)
