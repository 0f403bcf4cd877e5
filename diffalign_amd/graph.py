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


_STATES = {}


def _states_on(dev, idx: tuple) -> torch.Tensor:
    """The list of always-masked node classes on the device, uploaded once per (device, list): no host-to-device copy per call
    (one would also be illegal while a HIP graph is being captured)."""
    key = (str(dev), idx)
    if key not in _STATES:
        _STATES[key] = torch.tensor(idx, dtype=torch.int32, device=dev)
    return _STATES[key]


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
    states = _states_on(dev, tuple(node_idx))
    mx = torch.empty(B, n, dtype=torch.uint8, device=dev)
    me = torch.empty(B, n, dtype=torch.uint8, device=dev)
    am = orig.atom_map_numbers.contiguous() if orig.atom_map_numbers is not None else None
    L.check(L.lib().da_free_masks(L.ptr(X, torch.float32), L.ptr(L.as_u8(orig.node_mask)),
                                  L.ptr(am, torch.int64) if am is not None else L.ptr(None), C.c_int(B), C.c_int(n),
                                  C.c_int(kx), C.c_int(_MODES[mode]), C.c_int(atom_decoder.index('SuNo')),
                                  C.c_int(1 if include_supernode else 0), L.ptr(states), C.c_int(len(node_idx)),
                                  L.ptr(mx), L.ptr(me), L.stream()), 'da_free_masks')
    return mx.view(torch.bool), me.view(torch.bool)


def free_rows(node_masks) -> torch.Tensor:
    """Row numbers b n + i (int32, ascending) of the nodes whose class or whose edges may change at a reverse step (get_node_masks'
    pair).  Every pair with another endpoint keeps the condition's value (apply_mask, graph.py:411-470), so the denoiser's last layer
    has nothing to compute for the other rows (DiscreteDenoisingDiffusion.reverse_step last_rows=).  One host sync: once per chain."""
    need = node_masks[0].bool() | node_masks[1].bool()
    return torch.nonzero(need.reshape(-1)).squeeze(1).to(torch.int32).contiguous()


class LastLayersPlan:
    """What the last two layers of a fused reverse step compute (DiscreteDenoisingDiffusion.reverse_step last_rows=): the step reads the
    node logits of free nodes and the edge logits of pairs with both ends free, nothing else of the denoiser's output.
    rows / n_rows_dev / hint: the last layer's query rows (free_rows), optionally with the list's length in device memory and an estimate
    tiles_last: per query row b n + i, bit jt set = the last layer's E'[i][keys 32 jt ..] is wanted (i free for E and a key of that tile too)
    tiles_prev: the same for the layer before: its E' is the last layer's input, read for the rows of the list only
    share_last / share_prev: share of a full launch's algorithmic FLOPs those launches keep (bench.py's roofline; None: not computed)"""
    __slots__ = ('rows', 'n_rows_dev', 'hint', 'tiles_prev', 'tiles_last', 'share_prev', 'share_last')

    def __init__(self, rows, n_rows_dev=None, hint=None, tiles_prev=None, tiles_last=None, share_prev=None, share_last=None):
        self.rows, self.n_rows_dev, self.hint = rows, n_rows_dev, (int(rows.numel()) if hint is None else int(hint))
        self.tiles_prev, self.tiles_last, self.share_prev, self.share_last = tiles_prev, tiles_last, share_prev, share_last


def row_tile_masks(node_masks) -> Tuple[torch.Tensor, torch.Tensor]:
    """(tiles_prev, tiles_last) of LastLayersPlan for get_node_masks' pair: two (B n,) int32 tensors (no host sync).  n <= 1024."""
    mx, me = node_masks[0].bool(), node_masks[1].bool()
    B, n = me.shape
    nt = (n + 31) // 32
    assert nt <= 32, 'row tile masks: n <= 1024'
    key_tiles = torch.nn.functional.pad(me, (0, nt * 32 - n)).reshape(B, nt, 32).any(-1)                 # (B, nt): tile has a key free for E
    bits = (key_tiles.long() << torch.arange(nt, device=me.device)).sum(-1)                            # (B,)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits)                                           # as the int32 bit pattern
    tiles_last = torch.where(me, bits[:, None], torch.zeros_like(bits[:, None])).to(torch.int32).reshape(-1).contiguous()
    tiles_prev = torch.where(mx | me, -1, 0).to(torch.int32).reshape(-1).contiguous()
    return tiles_prev, tiles_last


def last_layers_plan(node_masks) -> LastLayersPlan:
    """LastLayersPlan of a chain from get_node_masks' pair: once per chain (host syncs for the list's length and the FLOP shares)."""
    mx, me = node_masks[0].bool(), node_masks[1].bool()
    B, n = me.shape
    rows = free_rows(node_masks)
    if n > 1024:
        return LastLayersPlan(rows)
    tiles_prev, tiles_last = row_tile_masks(node_masks)
    nt = (n + 31) // 32
    need = (mx | me).float()
    # algorithmic FLOPs per pair: the attention part (e_mul / e_add projection) is half, e_out + the feed-forward block the other half
    kt = torch.nn.functional.pad(me, (0, nt * 32 - n)).reshape(B, nt, 32).any(-1).float().sum(-1) / nt        # wanted key tiles per graph
    share_prev = float((need + (1 - need) * 0.5).mean())
    share_last = float((need * 0.5 + me.float() * 0.5 * kt[:, None]).mean())
    return LastLayersPlan(rows, None, None, tiles_prev, tiles_last, share_prev, share_last)


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


def inpaint_lists(node_idx, edge_idx, device):
    """Python index lists (one list per graph, graph.py:646-654) -> int32 device tensors nodes (Nn,2) rows (b,i), edges (Ne,3)
    rows (b,i,j); None when there is nothing to fix."""
    nodes = [(b, int(i)) for b, idx in enumerate(node_idx or []) for i in idx]
    edges = [(b, int(bond[0]), int(bond[1])) for b, bonds in enumerate(edge_idx or []) for bond in bonds]
    tn = torch.tensor(nodes, dtype=torch.int32, device=device).reshape(-1, 2) if nodes else None
    te = torch.tensor(edges, dtype=torch.int32, device=device).reshape(-1, 3) if edges else None
    return tn, te


def _inpaint(nodes, edges, B, n, kx=0, ke=0, mask_X=None, mask_E=None, z=None, orig=None):
    if nodes is None and edges is None:
        return
    zX = zE = oX = oE = None
    if z is not None:
        zX, zE, oX, oE = z.X, z.E, orig.X.contiguous().float(), orig.E.contiguous().float()
        assert zX.is_contiguous() and zE.is_contiguous() and zX.dtype == torch.float32 and zE.dtype == torch.float32
    L.check(L.lib().da_inpaint(L.ptr(nodes, torch.int32), 0 if nodes is None else nodes.shape[0], L.ptr(edges, torch.int32),
                               0 if edges is None else edges.shape[0], L.ptr(mask_X), L.ptr(mask_E), L.ptr(zX), L.ptr(zE), L.ptr(oX),
                               L.ptr(oE), B, n, kx, ke, L.stream()), 'da_inpaint')


def fix_nodes_and_edges_by_idx(data, node_idx, edge_idx):
    """graph.py:621-656: inpainting masks (B,n) / (B,n,n) bool from index lists (all False when None), set on the device by one
    scatter over the lists; no deepcopy / collapse of the batch."""
    B, n = data.X.shape[:2]
    dev = data.X.device
    mask_X = torch.zeros(B, n, dtype=torch.uint8, device=dev)
    mask_E = torch.zeros(B, n, n, dtype=torch.uint8, device=dev)
    nodes, edges = inpaint_lists(node_idx, edge_idx, dev)
    _inpaint(nodes, edges, B, n, mask_X=mask_X, mask_E=mask_E)
    return mask_X.view(torch.bool), mask_E.view(torch.bool)


def apply_inpainting(z_t, orig, nodes, edges):
    """diffusion/sampling.py:44,177: z_t.X[mask_X], z_t.E[mask_E] = orig.X[mask_X], orig.E[mask_E], in place, from the index
    tensors of inpaint_lists (work proportional to the lists, not to B n^2)."""
    B, n, kx = z_t.X.shape
    _inpaint(nodes, edges, B, n, kx, z_t.E.shape[-1], z=z_t, orig=orig)
    return z_t


def to_dense(data):
    """graph.py:189-210 on the device: `data` carries the PyG batch fields x (N,kx), edge_index (2,M), edge_attr (M,ke), batch (N,
    sorted), y and optionally mask_atom_mapping / mol_assignment (attributes or mapping keys).  The largest graph fixes n (one
    host sync, at the data-loading boundary as in the reference)."""
    get = (lambda k: data.get(k)) if isinstance(data, dict) else (lambda k: getattr(data, k, None))
    x, batch = get('x').contiguous().float(), get('batch').contiguous().long()
    dev = x.device
    ei, ea = get('edge_index').contiguous().long(), get('edge_attr').contiguous().float()
    am, mol = get('mask_atom_mapping'), get('mol_assignment')
    N, kx = x.shape
    M, ke = ea.shape
    B = int(batch[-1].item()) + 1 if N else 0
    n = int(torch.bincount(batch, minlength=B).max().item()) if N else 0
    X = torch.empty(B, n, kx, device=dev); E = torch.empty(B, n, n, ke, device=dev)
    nm = torch.empty(B, n, dtype=torch.uint8, device=dev)
    amd = torch.empty(B, n, dtype=torch.int64, device=dev) if am is not None else None
    mold = torch.empty(B, n, dtype=torch.int64, device=dev) if mol is not None else None
    ptr = torch.empty(B + 1, dtype=torch.int64, device=dev)
    L.check(L.lib().da_to_dense(L.ptr(x), L.ptr(batch, torch.int64), L.ptr(ei, torch.int64), L.ptr(ea),
                                L.ptr(am.contiguous().long() if am is not None else None), L.ptr(mol.contiguous().long() if mol is not None else None),
                                N, M, B, n, kx, ke, L.ptr(X), L.ptr(E), L.ptr(nm), L.ptr(amd), L.ptr(mold), L.ptr(ptr), L.stream()), 'da_to_dense')
    return PlaceHolder(X=X, E=E, y=get('y'), node_mask=nm.view(torch.bool), atom_map_numbers=amd, mol_assignments=mold)


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


def save_gen_rxn_smiles_to_file(filename, condition_idx, gen_rxns, true_rxn_str):
    """utils/graph.py:786-800, byte for byte: `(cond i) true_rxn:` then one tab-indented line per generated reaction; condition 0
    truncates the file, later conditions append.  `true_rxn_str`: the string mol.rxn_list_to_str(true_rcts, true_prods) gives (the
    chemistry side is the caller's: rdkit is not part of this package)."""
    with open(filename, 'w' if condition_idx == 0 else 'a') as f:
        f.write(f'(cond {condition_idx}) {true_rxn_str}:\n')
        f.writelines([f'\t{p}\n' for p in gen_rxns])


def save_samples_to_file_without_weighted_prob(filename, condition_idx, gen_rxns, true_rxn_strs):
    """utils/graph.py:772-784: per condition a header line and one line `(rxn string, [elbo, loss_t, loss_0, count])` per distinct
    sample.  gen_rxns: {key: [dict(rxn=str, elbo=, loss_t=, loss_0=, count=)]} in condition order; true_rxn_strs: one string each."""
    import os
    with open(filename, 'a' if os.path.exists(filename) else 'w') as f:
        for i, p in enumerate(gen_rxns):
            f.write(f'(cond {condition_idx + i}) {true_rxn_strs[i]}:\n')
            f.writelines([f'\t{(x["rxn"], [x["elbo"], x["loss_t"], x["loss_0"], x["count"]])}\n' for x in gen_rxns[p]])


# ---- host-side list logic of the scoring driver (utils/graph.py:160-187, 738-770; pure Python on reaction strings) ----------------

def get_unique_indices_from_reaction_list(gen_rxn_smiles):
    """utils/graph.py:160-187.  Reactions 'r1.r2>>p' are equal when their sorted reactant sets and their product are; returns
    (index of the first occurrence of every distinct reaction in increasing order, how often each occurs, per-sample flag
    'is a first occurrence').  The reference gets the same through np.unique on the normalised strings."""
    first, count = {}, {}
    for i, rxn in enumerate(gen_rxn_smiles):
        rct, prod = rxn.split('>>')[0], rxn.split('>>')[1]
        key = '.'.join(sorted(rct.split('.'))) + '>>' + prod
        if key not in first:
            first[key] = i
            count[key] = 0
        count[key] += 1
    keys = sorted(first, key=lambda k: first[k])
    unique_indices = [first[k] for k in keys]
    firsts = set(unique_indices)
    return unique_indices, [count[k] for k in keys], [i in firsts for i in range(len(gen_rxn_smiles))]


def split_reactions_to_reactants_and_products(reactions):
    """utils/graph.py:767-770: ([reactant, ...] per reaction, [product] per reaction)."""
    return [r.split('>>')[0].split('.') for r in reactions], [[r.split('>>')[1]] for r in reactions]


def calculate_top_k(cfg, elbo_sorted_rxns, true_rcts, true_prods):
    """utils/graph.py:738-765: per condition (the dict's insertion order) whether the true reaction -- as the SET of its reactant and
    product strings -- is among the first k candidates for every k of cfg.test.topks, and the reciprocal rank of its first match
    (0 when absent).  Returns (topk (bs, len(topks)) float, mrr (bs,) float)."""
    ks = list(cfg.test.topks)
    truth = [set(r) | set(p) for r, p in zip(true_rcts, true_prods)]
    topk = torch.zeros(len(elbo_sorted_rxns), len(ks))
    mrr = torch.zeros(len(elbo_sorted_rxns))
    for i, rows in enumerate(elbo_sorted_rxns.values()):
        cands = [set(row['rcts']) | set(row['prod']) for row in rows]
        hit = next((rank for rank, c in enumerate(cands, 1) if c == truth[i]), None)
        for j, k in enumerate(ks):
            topk[i, j] = float(hit is not None and hit <= k)
        if hit is not None:
            mrr[i] = 1.0 / hit
    return topk, mrr
