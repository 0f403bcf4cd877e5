"""Dense graph container with the reference's PlaceHolder interface (diffalign/utils/placeholder.py:21-179).
Masking runs in the HIP library (da_mask_dense / da_collapse)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields
from typing import Optional

import torch

from . import _lib as L


def encode_no_element(A: torch.Tensor) -> torch.Tensor:
    """Same contract as diffalign/utils/placeholder.py:9-18 (in place, returns A)."""
    assert A.dim() >= 3
    if A.shape[-1] == 0:
        return A
    flat = A.reshape(-1, A.shape[-1])
    ones = torch.ones(flat.shape[0], dtype=torch.bool, device=A.device)
    if not flat.is_contiguous() or flat.data_ptr() != A.data_ptr():
        raise L.DiffAlignHipError('encode_no_element needs a contiguous tensor')
    L.check(L.lib().da_mask_dense(L.ptr(flat, torch.float32), L.ptr(None), L.ptr(L.as_u8(ones)), C.c_int(1),
                                  C.c_int(flat.shape[0]), C.c_int(flat.shape[1]), C.c_int(0), L.stream()),
            'da_mask_dense')
    return A


@dataclass
class PlaceHolder:
    X: torch.Tensor
    E: torch.Tensor
    y: torch.Tensor
    node_mask: Optional[torch.Tensor] = None
    atom_map_numbers: Optional[torch.Tensor] = None
    mol_assignments: Optional[torch.Tensor] = None

    def _apply(self, fn):
        return PlaceHolder(**{f.name: fn(getattr(self, f.name)) if isinstance(getattr(self, f.name), torch.Tensor)
                              else None for f in fields(self)})

    def _apply_inplace(self, fn):
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                setattr(self, f.name, fn(v))
        return self

    def type_as(self, x):
        return self._apply_inplace(lambda t: t.type_as(x))

    def to_device(self, device):
        return self._apply_inplace(lambda t: t.to(device))

    def to_cpu(self):
        return self._apply_inplace(lambda t: t.detach().cpu())

    def flatten(self, start_dim, end_dim):
        return self._apply(lambda t: t.flatten(start_dim=start_dim, end_dim=end_dim))

    def select_subset(self, selection):
        return self._apply(lambda t: t.clone()[selection])

    def slice_by_idx(self, idx):
        return self._apply(lambda t: t.clone()[:idx])

    def subset_by_idx(self, start_idx, end_idx):
        return self._apply(lambda t: t[start_idx:end_idx].clone())

    def get_new_object(self, **kwargs):
        """New PlaceHolder with the given fields replaced; every tensor field cloned (placeholder.py:72-79)."""
        return PlaceHolder(**{f.name: kwargs[f.name].clone() if isinstance(kwargs.get(f.name), torch.Tensor)
                              else getattr(self, f.name).clone() if isinstance(getattr(self, f.name), torch.Tensor)
                              else None for f in fields(self)})

    def mask(self, node_mask=None, collapse=False):
        """placeholder.py:81-102, on the GPU."""
        if node_mask is None:
            node_mask = self.node_mask
        assert node_mask is not None, 'node_mask is None.'
        B, n = node_mask.shape
        nm = L.as_u8(node_mask)
        if collapse:
            X = self.X.contiguous().float(); E = self.E.contiguous().float()
            Xi = torch.empty(B, n, dtype=torch.int64, device=X.device)
            Ei = torch.empty(B, n, n, dtype=torch.int64, device=X.device)
            L.check(L.lib().da_collapse(L.ptr(X, torch.float32), L.ptr(E, torch.float32), L.ptr(nm), L.ptr(Xi), L.ptr(Ei),
                                        C.c_int(B), C.c_int(n), C.c_int(X.shape[-1]), C.c_int(E.shape[-1]), L.stream()),
                    'da_collapse')
            self.X, self.E = Xi, Ei
        else:
            # the reference multiplies out of place and then writes in place on the product
            self.X = self.X.float().clone() if self.X.dtype != torch.float32 or not self.X.is_contiguous() else self.X.clone()
            self.E = self.E.float().clone() if self.E.dtype != torch.float32 or not self.E.is_contiguous() else self.E.clone()
            self.X = self.X.contiguous(); self.E = self.E.contiguous()
            L.check(L.lib().da_mask_dense(L.ptr(self.X, torch.float32), L.ptr(self.E, torch.float32), L.ptr(nm), C.c_int(B),
                                          C.c_int(n), C.c_int(self.X.shape[-1]), C.c_int(self.E.shape[-1]), L.stream()),
                    'da_mask_dense')
        return self

    # ---- batch bookkeeping of the sampling / evaluation drivers (placeholder.py:104-152): plain tensor plumbing
    def to_numpy(self):
        return self._apply_inplace(lambda t: t.detach().cpu().numpy())

    def reshape_bs_n_samples(self, bs, n_samples, n):
        self.X, self.E = self.X.reshape(bs, n_samples, n), self.E.reshape(bs, n_samples, n, n)
        self.y = torch.empty((bs, n_samples))
        for name in ('node_mask', 'atom_map_numbers', 'mol_assignments'):
            setattr(self, name, getattr(self, name).reshape(bs, n_samples, n))

    def pad_nodes(self, pad_size):
        """Zero-pad the node axis (collapsed or one-hot X / E) by pad_size."""
        import torch.nn.functional as F
        self.X = F.pad(self.X, (0, pad_size) if self.X.ndim == 2 else (0, 0, 0, pad_size))
        self.E = F.pad(self.E, (0, pad_size, 0, pad_size) if self.E.ndim == 3 else (0, 0, 0, pad_size, 0, pad_size))
        for name in ('node_mask', 'atom_map_numbers', 'mol_assignments'):
            setattr(self, name, F.pad(getattr(self, name), (0, pad_size)))

    def cat_by_batchdim(self, placeh):
        for f in fields(self):
            setattr(self, f.name, torch.cat((getattr(self, f.name), getattr(placeh, f.name)), dim=0))

    def cat_by_batchdim_with_padding(self, placeh):
        small, big = (placeh, self) if self.X.shape[1] > placeh.X.shape[1] else (self, placeh)
        small.pad_nodes(big.X.shape[1] - small.X.shape[1])
        big.cat_by_batchdim(small)
        return big

    def select_by_batch_idx(self, idx):
        return PlaceHolder(**{f.name: getattr(self, f.name)[idx:idx + 1].clone() if isinstance(getattr(self, f.name), torch.Tensor) else None
                              for f in fields(self)})

    def select_by_batch_and_sample_idx(self, bs, n_samples, batch_idx, sample_idx):
        assert self.X.ndim == 2 and self.E.ndim == 3, f'expected collapsed X (bs, n) / E (bs, n, n), got {self.X.shape} / {self.E.shape}'
        pick = lambda t, *tail: t.reshape(bs, n_samples, *tail)[batch_idx:batch_idx + 1, sample_idx]
        n = self.X.shape[1]
        return PlaceHolder(X=pick(self.X, n), E=pick(self.E, n, n), y=pick(self.y, -1), node_mask=pick(self.node_mask, n),
                           atom_map_numbers=pick(self.atom_map_numbers, n), mol_assignments=pick(self.mol_assignments, n))

    def to_sparse(self):
        """The fields of the torch_geometric `Data` objects PlaceHolder.pyg() builds (utils/placeholder.py:165-179), one dict per
        graph, with the reference's dtypes: x uint8 class indices, edge_index int16 (2,M) and edge_attr uint8 (M) of the non-zero
        entries of the collapsed E in row-major order (torch_geometric.utils.dense_to_sparse), y / node_mask / mask_atom_mapping /
        mol_assignment uint8.  torch_geometric is not a dependency here: wrapping each dict as Data(**d) and Batch.from_data_list
        gives the reference's object (what its gzip-pickled `samples_*.gz` files hold)."""
        X = self.X if self.X.dim() == 2 else self.X.argmax(-1)
        E = self.E if self.E.dim() == 3 else self.E.argmax(-1)
        out = []
        for i in range(X.shape[0]):
            idx = E[i].nonzero().t().contiguous()
            out.append(dict(x=X[i].to(torch.uint8), edge_index=idx.to(torch.int16), edge_attr=E[i][idx[0], idx[1]].to(torch.uint8),
                            y=self.y.to(torch.uint8) if self.y is not None else None, node_mask=self.node_mask[i].to(torch.uint8),
                            mask_atom_mapping=self.atom_map_numbers[i].to(torch.uint8), mol_assignment=self.mol_assignments[i].to(torch.uint8)))
        return out

    def serialize(self):
        return {f.name: (getattr(self, f.name).detach().cpu().numpy().tolist() if getattr(self, f.name) is not None else None) for f in fields(self)}

    @classmethod
    def deserialize(cls, data: dict) -> 'PlaceHolder':
        return cls(**{k: torch.tensor(v) if v is not None else None for k, v in data.items()})
