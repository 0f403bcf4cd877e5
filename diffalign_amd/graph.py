"""Masking / dense-graph utilities with the reference's names (diffalign/utils/graph.py), running on the GPU
through the HIP library.  The (B*n)^2 outer product of graph.py:612 and the per-step deepcopy of
graph.py:636 are gone: masks are per-node uint8 vectors and are expanded inside the kernels."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _lib as L
from .placeholder import PlaceHolder, encode_no_element  # noqa: F401  (re-exported like the reference does)

_MODES = {'product': 0, 'sn': 1, 'product_and_sn': 2, 'reactant': 3, 'reactant_and_sn': 4, 'atom_mapping': 5}


def get_index_from_states(atom_decoder, bond_decoder, node_states_to_mask, edge_states_to_mask, device=None):
    """graph.py:678-699 ('none' is always among the fixed node states); returns python lists."""
    node_states = list(node_states_to_mask or [])
    if 'none' not in node_states:
        node_states.append('none')
    missing = [a for a in node_states if a not in atom_decoder]
    assert not missing, f'node_states_to_mask contains states not in atom_decoder: {missing}'
    node_idx = [atom_decoder.index(a) for a in node_states]
    edge_idx = None
    if edge_states_to_mask is not None:
        missing = [a for a in edge_states_to_mask if a not in bond_decoder]
        assert not missing, f'edge_states_to_mask contains states not in bond_decoder: {missing}'
        edge_idx = [bond_decoder.index(a) for a in edge_states_to_mask]
    return node_idx, edge_idx


def get_node_masks(orig: PlaceHolder, atom_decoder, bond_decoder, mask_nodes, mask_edges, node_states_to_mask=(),
                   edge_states_to_mask=(), include_supernode=True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-node free masks (bool (B,n) each): closed form of graph.py:462-534."""
    mode = mask_nodes if mask_nodes in _MODES else mask_edges
    if mode not in _MODES:
        raise ValueError(f'mask mode {mask_nodes!r}/{mask_edges!r} is not supported (the reference raises '
                         'UnboundLocalError at graph.py:524 for it)')
    node_idx, _ = get_index_from_states(atom_decoder, bond_decoder, list(node_states_to_mask), None)
    B, n, kx = orig.X.shape
    dev = orig.X.device
    X = orig.X.contiguous().float()
    states = torch.tensor(node_idx, dtype=torch.int32, device=dev)
    mx = torch.empty(B, n, dtype=torch.uint8, device=dev)
    me = torch.empty(B, n, dtype=torch.uint8, device=dev)
    am = orig.atom_map_numbers.contiguous() if orig.atom_map_numbers is not None else None
    L.check(L.lib().da_free_masks(L.ptr(X, torch.float32), L.ptr(L.as_u8(orig.node_mask)),
                                  L.ptr(am, torch.int64) if am is not None else L.ptr(None), C.c_int(B), C.c_int(n),
                                  C.c_int(kx), C.c_int(_MODES[mode]), C.c_int(atom_decoder.index('SuNo')),
                                  C.c_int(1 if include_supernode else 0), L.ptr(states), C.c_int(len(node_idx)),
                                  L.ptr(mx), L.ptr(me), L.stream()), 'da_free_masks')
    return mx.view(torch.bool), me.view(torch.bool)


def from_mask_to_maskX_maskE(mask_nodes, mask_edges, shape_X, shape_E):
    """graph.py:607-619 (expanded views, built only when a caller asks for the dense masks)."""
    mask_X = mask_nodes.unsqueeze(-1).expand(-1, -1, shape_X[-1])
    mask_E = (mask_edges[:, :, None] & mask_edges[:, None, :]).unsqueeze(-1).expand(-1, -1, -1, shape_E[-1])
    return mask_X, mask_E


def get_mask(orig, atom_decoder, bond_decoder, mask_nodes, mask_edges, node_states_to_mask, edge_states_to_mask,
             include_supernode, return_mask_nodes=False):
    mx, me = get_node_masks(orig, atom_decoder, bond_decoder, mask_nodes, mask_edges, node_states_to_mask,
                            edge_states_to_mask, include_supernode)
    if return_mask_nodes:
        return mx
    return from_mask_to_maskX_maskE(mx, me, orig.X.shape, orig.E.shape)


def apply_mask(orig, z_t, atom_decoder, bond_decoder, mask_nodes=None, mask_edges=None, node_states_to_mask=(),
               edge_states_to_mask=(), as_logits=False, return_masks=False, include_supernode=True, node_masks=None):
    """graph.py:411-460.  `node_masks=(mask_x, mask_e)` lets a loop reuse masks computed once."""
    if node_masks is None:
        node_masks = get_node_masks(orig, atom_decoder, bond_decoder, mask_nodes, mask_edges, node_states_to_mask,
                                    edge_states_to_mask, include_supernode)
    mx, me = node_masks
    z = z_t.get_new_object()
    B, n, kx = z.X.shape
    ke = z.E.shape[-1]
    z.X = z.X.contiguous().float(); z.E = z.E.contiguous().float()
    oX = orig.X.contiguous().float(); oE = orig.E.contiguous().float()
    L.check(L.lib().da_apply_mask(L.ptr(z.X), L.ptr(z.E), L.ptr(oX), L.ptr(oE), L.ptr(L.as_u8(mx)), L.ptr(L.as_u8(me)),
                                  C.c_int(B), C.c_int(n), C.c_int(kx), C.c_int(ke), C.c_int(1 if as_logits else 0),
                                  L.stream()), 'da_apply_mask')
    if return_masks:
        mX, mE = from_mask_to_maskX_maskE(mx, me, orig.X.shape, orig.E.shape)
        return z, mX, mE
    return z


def fix_nodes_and_edges_by_idx(data, node_idx, edge_idx):
    """graph.py:621-656: inpainting masks from index lists (all False when None); no deepcopy/collapse."""
    B, n = data.X.shape[:2]
    mask_X = torch.zeros(B, n, dtype=torch.bool, device=data.X.device)
    mask_E = torch.zeros(B, n, n, dtype=torch.bool, device=data.X.device)
    if node_idx is not None:
        for b in range(B):
            mask_X[b, node_idx[b]] = True
    if edge_idx is not None:
        for b in range(B):
            for bond in edge_idx[b]:
                mask_E[b, bond[0], bond[1]] = True
                mask_E[b, bond[1], bond[0]] = True
    return mask_X, mask_E


def duplicate_data(dense_data, n_samples=1, mask_=True, get_discrete_data=False):
    """graph.py:130-158."""
    if mask_:
        dense_data = dense_data.mask(dense_data.node_mask)
    for name in ('X', 'E', 'node_mask', 'y', 'atom_map_numbers', 'mol_assignments'):
        v = getattr(dense_data, name)
        if v is not None:
            setattr(dense_data, name, v.repeat_interleave(n_samples, dim=0))
    if get_discrete_data:
        disc = dense_data.get_new_object().mask(dense_data.node_mask, collapse=True)
        return dense_data, disc
    return dense_data
