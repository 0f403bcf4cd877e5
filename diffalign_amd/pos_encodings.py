"""Positional encodings shared between mapped reactant / product atoms, on the device.

Host-side mirror of `PositionalEmbedding` (neuralnet/transformer_model_with_y.py:213-503): same constructor and method
names, direction 'retro' (the reference's 'forward' branches do not run: `torch.scatter(indx=...)`, :309, :445).  The
reference moves atom_map_numbers / mol_assignments to the CPU and loops over the batch in Python; here the product index,
the sinusoidal table, the row gathers and the product-block Laplacian are HIP kernels (csrc/k_posenc.hip) and the
eigendecomposition is one batched `torch.linalg.eigh` on the device (fp32, as `laplacian_eigenvectors_gpu`, :197-201).

The Laplacian encodings are eigenvectors: defined up to a sign per vector (the reference flips signs at random, :169, :208)
and up to a basis inside degenerate eigenspaces -- in particular the n - P isolated non-product nodes give an (n - P)-fold
eigenvalue 1 whose eigenvectors ARPACK / LAPACK return in an arbitrary basis; where the product block has fewer than k + 1
eigenvalues below that cluster, the reference's remaining columns are such arbitrary vectors (zero on product rows unless
the block itself has the eigenvalue 1).  This build takes the same columns 1..k of the ascending dense decomposition."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L


def _i(v):
    return C.c_int(int(v))


class SinusoidalPosEmb(nn.Module):
    """transformer_model_with_y.py:114-131."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self._freq = {}

    def freq(self, dev):
        if dev not in self._freq:
            half_dim = self.dim // 2
            emb = math.log(10000) / (half_dim - 1)
            self._freq[dev] = torch.exp(torch.arange(half_dim) * -emb).float().to(dev)       # :126-128, evaluated as torch does
        return self._freq[dev]

    def forward(self, x):
        x = x.reshape(-1).float().contiguous()
        out = torch.empty(x.numel(), self.dim, device=x.device)
        L.check(L.lib().da_pos_sinusoidal(L.ptr(x, torch.float32), L.ptr(self.freq(x.device)), L.ptr(out), C.c_longlong(x.numel()),
                                          _i(self.dim), L.stream()), 'da_pos_sinusoidal')
        return out


class PositionalEmbedding(nn.Module):
    def __init__(self, dim, pos_emb_permutations=0):
        super().__init__()
        self.dim = dim
        self.pos_emb = SinusoidalPosEmb(dim=dim)
        self.pos_emb_permutations = pos_emb_permutations

    # -- product index of every node's atom-map number (:400-412, :259-275)
    @staticmethod
    def product_positions(atom_map_numbers, molecule_assignments):
        B, n = atom_map_numbers.shape
        dev = atom_map_numbers.device
        am = atom_map_numbers.long().contiguous(); mol = molecule_assignments.long().contiguous()
        pos = torch.empty(B, n, dtype=torch.int32, device=dev)
        prod_node = torch.empty(B, n + 1, dtype=torch.int32, device=dev)
        n_prod = torch.empty(B, dtype=torch.int32, device=dev)
        L.check(L.lib().da_pos_product_index(L.ptr(am, torch.int64), L.ptr(mol, torch.int64), L.ptr(pos), L.ptr(prod_node), L.ptr(n_prod),
                                             _i(B), _i(n), L.stream()), 'da_pos_product_index')
        return pos, prod_node, n_prod

    def matched_positional_encodings_sinusoidal(self, atom_map_numbers, molecule_assignments, direction='retro'):
        """:389-417: sinusoidal embedding of the rank, inside the product, of the atom carrying the same atom-map number."""
        assert direction == 'retro', "only direction='retro' is built (the reference's 'forward' branch does not run)"
        assert molecule_assignments is not None, 'Need to provide molecule assigments'
        B, n = atom_map_numbers.shape
        pos, _, _ = self.product_positions(atom_map_numbers, molecule_assignments)
        return self.pos_emb(pos.float()).reshape(B, n, self.dim)

    def matched_positional_encodings_infoleak(self, atom_map_numbers, molecule_assignments, direction='retro'):
        """:451-490 with pos_emb_permutations = 0: sinusoidal embedding of the atom-map number itself."""
        B, n = atom_map_numbers.shape
        if self.pos_emb_permutations != 0:
            # :458-485: every graph's numbers 1..max are relabelled by one of `pos_emb_permutations` fixed shuffles (or, for -1, a fresh
            # one).  The shuffles come from Python's / numpy's host generators exactly as in the reference (same calls, same order), so
            # a seeded run relabels alike; the table goes to the device once and the relabelling is one gather.
            import random
            import numpy as np
            max_am = atom_map_numbers.max(-1).values.cpu().int()
            perms = [list(range(1, int(m) + 1)) for m in max_am]
            if self.pos_emb_permutations > 0:
                perm_numbers = np.random.randint(0, self.pos_emb_permutations, B)
                for idx, pn in enumerate(perm_numbers):
                    random.Random(pn).shuffle(perms[idx])
            else:
                for idx in range(B):
                    random.shuffle(perms[idx])
            width = int(max(max_am))
            table = torch.tensor([[0] + p + (width - len(p)) * [0] for p in perms], device=atom_map_numbers.device)
            atom_map_numbers = table.gather(1, atom_map_numbers.long())
        return self.pos_emb(atom_map_numbers.float()).reshape(B, n, self.dim)

    def matched_positional_encodings_gaussian(self, atom_map_numbers, molecule_assignments, noise: Optional[torch.Tensor] = None):
        """:492-503: N(0,1) rows indexed by atom-map number, row 0 zero.  `noise` (B,n,dim) stands for the reference's randn."""
        B, n = atom_map_numbers.shape
        dev = atom_map_numbers.device
        if noise is None:
            noise = torch.randn(B, n, self.dim, device=dev)
        noise = noise.to(dev).float().clone()
        noise[:, 0, :] = 0
        am = atom_map_numbers.long().contiguous()
        out = torch.empty(B, n, self.dim, device=dev)
        L.check(L.lib().da_pos_gather_rows(L.ptr(noise.contiguous(), torch.float32), L.ptr(am, torch.int64), L.ptr(out), _i(B), _i(n), _i(n),
                                           _i(self.dim), L.stream()), 'da_pos_gather_rows')
        return out

    def matched_positional_encodings_laplacian(self, E, atom_map_numbers, molecule_assignments, max_eigenvectors, direction='retro',
                                               signs: Optional[torch.Tensor] = None):
        """:221-287 (and the _gpu variant :314-387): the k = min(max_eigenvectors, P) lowest non-trivial eigenvectors of the
        symmetric-normalised Laplacian of the product block, shared between mapped atoms, zero padded to dim.
        E (B,n,n) bond classes (not one-hot).  `signs` (B,max_eigenvectors) of +-1 stands for the reference's random flips."""
        assert direction == 'retro', "only direction='retro' is built"
        assert molecule_assignments is not None, 'Need to provide molecule assigments'
        assert E.dim() == 3 and E.shape[1] == E.shape[2], 'E should not be in one-hot format'
        B, n = atom_map_numbers.shape
        dev = E.device
        pos, prod_node, n_prod = self.product_positions(atom_map_numbers, molecule_assignments)
        Lap = torch.empty(B, n, n, device=dev)
        L.check(L.lib().da_pos_laplacian(L.ptr(E.long().contiguous(), torch.int64), L.ptr(molecule_assignments.long().contiguous(), torch.int64),
                                         L.ptr(Lap), _i(B), _i(n), L.stream()), 'da_pos_laplacian')
        _, evecs = torch.linalg.eigh(Lap)                                                    # :201 (ascending)
        s = evecs.stride()
        assert s[0] == n * n
        if signs is None:
            signs = (-1 + 2 * torch.randint(0, 2, (B, max_eigenvectors), device=dev)).float()   # :169, :208
        out = torch.empty(B, n, self.dim, device=dev)
        L.check(L.lib().da_pos_lap_gather(C.c_void_p(evecs.data_ptr()), C.c_longlong(s[1]), C.c_longlong(s[2]), L.ptr(pos), L.ptr(prod_node),
                                          L.ptr(n_prod), L.ptr(signs.float().contiguous()), L.ptr(out), _i(B), _i(n), _i(self.dim),
                                          _i(max_eigenvectors), L.stream()), 'da_pos_lap_gather')
        return out

    matched_positional_encodings_laplacian_gpu = matched_positional_encodings_laplacian
