"""fp32 training forward of the denoiser under torch.autograd.

Same arithmetic as GraphTransformerWithY[AtomMapPosEmb].forward (neuralnet/transformer_model_with_y.py:74-111, 620-713) and
XEyTransformerLayer / NodeEdgeBlock (neuralnet/layers.py:204-361) in train mode, laid out for a backward pass:
  * every nn.Linear is a library GEMM under autograd, except that the weight gradients of the per-pair Linears (a (out,in) result
    reduced over B n n rows) come from da_linear_wgrad (autograd_ops.WideLinear);
  * the per-pair attention core, the LayerNorms, the pooled statistics and the hidden-feature mask are HIP autograd.Functions
    (diffalign_amd/autograd_ops.py);
  * masks, residual adds, ReLU, dropout, the symmetrisation and the product->reactant skip gather are torch elementwise glue.
The fused bf16 inference path (engine.model_forward) is not differentiable; training batches are small (8-16 graphs, n <= 140),
so this path materialises the 256-wide per-pair tensors in fp32 like the reference does (which runs it under fp16 autocast).
Parameters without a gradient in the reference (mlp_out_y, self_attn.y_y, skip_scaling_3: transformer_model_with_y.py:673-681,
layers.py:277-280) get none here either."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from .autograd_ops import AttnCore, AddLayerNorm, PoolStats, MaskDense, qscale_of, wide_linear


def _node_edge_block(a, X, E, y, nm_u8, xm, em):
    """layers.py:204-284."""
    Q, K, V = a.q(X) * xm, a.k(X) * xm, a.v(X) * xm
    Z, wV = AttnCore.apply(Q, K, V, wide_linear(a.e_mul, E), wide_linear(a.e_add, E), nm_u8, a.y_e_add(y), a.y_e_mul(y), qscale_of(a.dx, a.n_head))
    newE = wide_linear(a.e_out, Z) * em
    newX = a.x_out(a.y_x_add(y)[:, None, :] + (a.y_x_mul(y)[:, None, :] + 1) * wV) * xm
    new_y = a.y_out(y + a.x_y.lin(PoolStats.apply(X)) + a.e_y.lin(PoolStats.apply(E)))
    return newX, newE, new_y


def _node_edge_block_improved(a, X, E, y, nm_u8, xm, em):
    """layers.py:87-164 NodeEdgeBlockImproved: the edge update takes Y without the y FiLM, and attention is ONE softmax per head over the
    head's summed scores (the plain block has one per channel).  Off in both align_* experiments (configs/neuralnet/transformer.yaml:15);
    written in torch ops over the library GEMMs -- it has no fused HIP form, inference runs this same code without a graph."""
    B, n, dx = X.shape
    H, df = a.n_head, a.df
    Q, K, V = a.q(X) * xm, a.k(X) * xm, a.v(X) * xm
    Y = Q[:, :, None, :] * K[:, None, :, :] * qscale_of(a.dx, a.n_head)                        # (B,i,j,dx)
    Y = Y * (wide_linear(a.e_mul, E) * em + 1) + wide_linear(a.e_add, E) * em                 # :115-122
    newE = wide_linear(a.e_out, Y) * em                                                       # :123-125
    attn = torch.softmax(Y.reshape(B, n, n, H, df).sum(-1), dim=2)                            # :128-130 (over keys, no mask)
    wV = torch.einsum('bijh,bjhf->bihf', attn, V.reshape(B, n, H, df)).reshape(B, n, dx)      # :132-141
    newX = a.x_out(a.y_x_add(y)[:, None, :] + (a.y_x_mul(y)[:, None, :] + 1) * wV) * xm       # :148-153
    new_y = a.y_out(y + a.x_y.lin(PoolStats.apply(X)) + a.e_y.lin(PoolStats.apply(E)))       # :157-161
    return newX, newE, new_y


def _layer(layer, X, E, y, nm_u8, xm, em, training):
    """layers.py:330-361 (post-LN residual blocks; dropout only in train mode)."""
    drop = (lambda t: F.dropout(t, layer.dropout_p, True)) if training and layer.dropout_p > 0 else (lambda t: t)
    block = _node_edge_block_improved if layer.self_attn.improved else _node_edge_block
    nX, nE, ny = block(layer.self_attn, X, E, y, nm_u8, xm, em)
    X = AddLayerNorm.apply(X, drop(nX), layer.normX1.weight, layer.normX1.bias)
    E = AddLayerNorm.apply(E, drop(nE), layer.normE1.weight, layer.normE1.bias)
    y = AddLayerNorm.apply(y, drop(ny), layer.norm_y1.weight, layer.norm_y1.bias)
    X = AddLayerNorm.apply(X, drop(layer.linX2(drop(F.relu(layer.linX1(X))))), layer.normX2.weight, layer.normX2.bias)
    E = AddLayerNorm.apply(E, drop(wide_linear(layer.linE2, drop(F.relu(wide_linear(layer.linE1, E))))), layer.normE2.weight, layer.normE2.bias)
    y = AddLayerNorm.apply(y, drop(layer.lin_y2(drop(F.relu(layer.lin_y1(y))))), layer.norm_y2.weight, layer.norm_y2.bias)
    return X, E, y


def align_index(atom_map_numbers, mol_assignments, alignment_type='correct'):
    """src (B,n) int32: for every reactant-side node with an atom-map number its product node, else -1.
    'correct' (transformer_model_with_y.py:540-552): the product node carrying the same number (da_align_index).
    'old' (:526-538): create_permutation_matrix_torch on the two index lists minus their minima -- both increasing, so the matrix is
    the identity: the k-th mapped reactant-side node (in node order) takes the k-th mapped product node, whatever their numbers."""
    B, n = atom_map_numbers.shape
    dev = atom_map_numbers.device
    if alignment_type == 'correct':
        src = torch.empty(B, n, dtype=torch.int32, device=dev)
        L.check(L.lib().da_align_index(L.ptr(atom_map_numbers.contiguous(), torch.int64), L.ptr(mol_assignments.contiguous(), torch.int64),
                                       L.ptr(src), B, n, L.stream()), 'da_align_index')
        return src
    if alignment_type != 'old':
        raise ValueError(f'Alignment type not set correctly ({alignment_type}). Should be old or correct')
    is_prod = mol_assignments == mol_assignments.max(-1, keepdim=True).values
    mapped = atom_map_numbers > 0
    rct, prod = mapped & ~is_prod, mapped & is_prod
    ar = torch.arange(n, device=dev).expand(B, n)
    # node indices of the mapped product nodes in increasing order, padded with n
    prod_sorted = torch.where(prod, ar, torch.full_like(ar, n)).sort(-1).values
    rank = rct.long().cumsum(-1) - 1
    src = torch.where(rct, prod_sorted.gather(1, rank.clamp_min(0)), torch.full_like(ar, -1))
    return torch.where(src == n, torch.full_like(src, -1), src).int()          # (the reference asserts equal counts, :523)


def aligned_product_features(X, E, atom_map_numbers, mol_assignments, kx, ke, alignment_type='correct'):
    """transformer_model_with_y.py:506-555 (alignment 'correct') + :633-645 as gathers: for every reactant-side node r with an atom-map
    number, the input features of the product node p(r) carrying the same number.  Returns (X_al (B,n,kx): X[p(r), :kx] on mapped
    rows, else 0;  E_al (B,n,n,ke): E[p(r1), p(r2), :ke] on mapped pairs, else one-hot class 0;  pair (B,n,n,1) bool)."""
    B, n = X.shape[:2]
    src = align_index(atom_map_numbers, mol_assignments, alignment_type)
    sel = src >= 0
    idx = src.clamp_min(0).long()
    X_al = torch.gather(X[..., :kx], 1, idx[:, :, None].expand(-1, -1, kx)) * sel[:, :, None]
    pair = (sel[:, :, None] & sel[:, None, :])[..., None]
    bi = torch.arange(B, device=X.device)[:, None, None]
    first = (torch.arange(ke, device=X.device) == 0).float()          # class 0; built on the device (graph-capturable)
    E_al = E[bi, idx[:, :, None], idx[:, None, :]][..., :ke] * pair + first * (~pair)
    return X_al, E_al, pair


def forward_train(m, X, E, y, node_mask, atom_map_numbers=None, pos_encodings=None, mol_assignments=None):
    """Differentiable forward of the denoiser module `m`; returns (X logits, E logits, y) like the module's forward."""
    if not X.is_cuda:
        raise L.DiffAlignHipError('denoiser inputs must be on a HIP device (no CPU fallback)')
    X, E, y = X.float(), E.float(), y.float()
    B, n = X.shape[:2]
    X_al = E_al = None
    if getattr(m, 'input_alignment', False):                             # :630-647: the aligned product features as extra input channels
        X_al, E_al, _ = aligned_product_features(X, E, atom_map_numbers, mol_assignments, m.out_dim_X, m.out_dim_E, m.alignment_type)
        X, E = torch.cat([X, X_al], -1), torch.cat([E, E_al], -1)
    nm_u8 = L.as_u8(node_mask)
    xm = node_mask[:, :, None].float()
    em = xm[:, :, None, :] * xm[:, None, :, :]
    Xh, yh = m.mlp_in_X(X), m.mlp_in_y(y)
    Eh = F.relu(wide_linear(m.mlp_in_E[2], F.relu(m.mlp_in_E[0](E))))
    Eh = (Eh + Eh.transpose(1, 2)) / 2
    Xh, Eh = MaskDense.apply(Xh, Eh, nm_u8)
    if pos_encodings is not None:
        Xh = Xh + pos_encodings.float()
    for layer in m.tf_layers:
        Xh, Eh, yh = _layer(layer, Xh, Eh, yh, nm_u8, xm, em, m.training)
    kx, ke, ky = m.out_dim_X, m.out_dim_E, m.out_dim_y
    Xo = m.mlp_out_X(Xh) + X[..., :kx]
    Eo = (m.mlp_out_E[2](F.relu(wide_linear(m.mlp_out_E[0], Eh))) + E[..., :ke]) * (~torch.eye(n, dtype=torch.bool, device=X.device))[None, :, :, None]
    Eo = 0.5 * (Eo + Eo.transpose(1, 2))
    yo = yh[..., :ky] + y[..., :ky]
    if getattr(m, 'p_to_r_skip_connection', False):
        # transformer_model_with_y.py:686-704 as a gather: mapped reactant-side nodes / pairs take s1 times the product's input features,
        # every other pair s2 on class 0; with input_alignment the reference reuses the concatenated tensors and scales ALL of the edge
        # part by s2 (:702-704)
        if X_al is None:
            X_al, E_al, pair = aligned_product_features(X, E, atom_map_numbers, mol_assignments, kx, ke, m.alignment_type)
            Eo = Eo + torch.where(pair, m.skip_scaling * E_al, m.skip_scaling_2 * E_al)
        else:
            Eo = Eo + m.skip_scaling_2 * E_al
        Xo = Xo + m.skip_scaling * X_al
    return Xo, Eo, yo
