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

import os as _os

import torch
import torch.nn.functional as F

from . import _lib as L
from .autograd_ops import (AttnCore, AttnCoreBF16, AddLayerNorm, PoolStats, MaskDense, ResidualLN, ReluDropout, FUSED_WIDTHS, qscale_of,
                           wide_linear)


# The node (B n rows) and y (B rows) sides of a layer are ~150 launches of a few microseconds each, forward plus backward, beside a
# dozen large per-pair kernels: 1870 of a step's 2340 launches and 12 of its 38 ms at B = 16, n = 140 when they run in line.  They
# share no data with the pair side except at three points per layer (q/k/v + FiLM vectors in, wV and the pooled edge statistics out),
# so they run on a second stream; autograd replays every backward node on the stream of its forward, which gives the backward pass
# the same overlap.  Tensors that cross are registered with the allocator (record_stream).  DA_TRAIN_SIDE_STREAM=0: one stream.
SIDE_STREAM = _os.environ.get('DA_TRAIN_SIDE_STREAM', '1') != '0'
_SIDE = {}


def _side_stream(dev):
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class _Lanes:
    """The two streams of one forward_train call: `pair` (the caller's) and `node` (the side stream, or the same one)."""

    def __init__(self, dev, enabled):
        self.pair = torch.cuda.current_stream(dev)
        self.node = _side_stream(dev) if enabled else self.pair
        self.split = self.node is not self.pair

    def on_node(self):
        return torch.cuda.stream(self.node)

    def to_pair(self, *ts):
        """results of the node stream, about to be read on the pair stream"""
        if self.split:
            self.pair.wait_stream(self.node)
            for t in ts:
                if t is not None: t.record_stream(self.pair)
        return ts

    def to_node(self, *ts):
        if self.split:
            self.node.wait_stream(self.pair)
            for t in ts:
                if t is not None: t.record_stream(self.node)
        return ts


def _node_edge_block(a, X, E, y, nm_u8, xm, em, Eb=None, lanes=None, fused=False):
    """layers.py:204-284.  `Eb`: E in bf16 (mixed precision): the 256-wide per-pair tensors E1, E2, Z and their gradients are bf16
    in memory, the three projections bf16 GEMMs with fp32 accumulation; the node and y sides stay fp32."""
    with lanes.on_node():
        Q, K, V = a.q(X) * xm, a.k(X) * xm, a.v(X) * xm
        yea, yem = a.y_e_add(y), a.y_e_mul(y)
        pX = PoolStats.apply(X)
    pE = PoolStats.apply(E)
    bf = Eb is not None and a.dx % 4 == 0 and 256 % (a.dx // 4) == 0
    src = Eb if bf else E
    E1, E2 = wide_linear(a.e_mul, src), wide_linear(a.e_add, src)
    lanes.to_pair(Q, K, V, yea, yem)
    Z, wV = (AttnCoreBF16 if bf else AttnCore).apply(Q, K, V, E1, E2, nm_u8, yea, yem, qscale_of(a.dx, a.n_head))
    lanes.to_node(wV, pE)
    newE = wide_linear(a.e_out, Z)
    if not fused:                     # fused: the mask (and the widening) happen inside the residual block's kernel
        newE = (newE.float() if bf else newE) * em
    with lanes.on_node():
        newX = a.x_out(a.y_x_add(y)[:, None, :] + (a.y_x_mul(y)[:, None, :] + 1) * wV) * xm
        new_y = a.y_out(y + a.x_y.lin(pX) + a.e_y.lin(pE))
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


# Fused residual blocks (csrc/k_train_fused.hip: mask + dropout + add + LayerNorm (+ bf16 copy for the next GEMM) in one kernel each
# way, ReLU + dropout in place).  DA_TRAIN_FUSED=0 / module.train_fused = False: the same blocks as torch elementwise passes.
FUSED = _os.environ.get('DA_TRAIN_FUSED', '1') != '0'
_SEEDS = {}


def _dropout_seed(dev):
    """A fresh device-resident seed word for this forward pass (advanced by a device op, so a captured step advances it per replay)."""
    dev = torch.device(dev)
    st = _SEEDS.get(dev)
    if st is None:
        st = _SEEDS[dev] = torch.full((1,), torch.initial_seed() & 0x7fffffffffff, dtype=torch.int64, device=dev)
    st.add_(0x9e3779b97f4a7c)
    return st.clone()


def _layer(layer, X, E, y, nm_u8, xm, em, training, bf16=False, lanes=None, Eb=None, fuse=None):
    """layers.py:330-361 (post-LN residual blocks; dropout only in train mode).  X and y live on lanes.node, E on lanes.pair.
    `fuse` = (seed tensor, salt base) selects the fused residual blocks; then `Eb` carries E in bf16 between layers (mixed precision)
    and the layer returns (X, E, y, Eb)."""
    p = layer.dropout_p if training else 0.0
    drop = (lambda t: F.dropout(t, p, True)) if p > 0 else (lambda t: t)
    B, n = X.shape[:2]
    if fuse is not None:
        seed, salt = fuse
        if bf16 and Eb is None:
            Eb = E.to(torch.bfloat16)
        nX, nE, ny = _node_edge_block(layer.self_attn, X, E, y, nm_u8, xm, em, Eb if bf16 else None, lanes, fused=True)
        with lanes.on_node():
            X, _ = ResidualLN.apply(X, nX, nm_u8, 0, n, layer.normX1.weight, layer.normX1.bias, seed, salt + 0, p, False)   # nX is masked already
            y, _ = ResidualLN.apply(y, ny, nm_u8, 0, n, layer.norm_y1.weight, layer.norm_y1.bias, seed, salt + 1, p, False)
            h = ReluDropout.apply(layer.linX1(X), seed, salt + 2, p)
            X, _ = ResidualLN.apply(X, layer.linX2(h), nm_u8, 0, n, layer.normX2.weight, layer.normX2.bias, seed, salt + 3, p, False)
            h = ReluDropout.apply(layer.lin_y1(y), seed, salt + 4, p)
            y, _ = ResidualLN.apply(y, layer.lin_y2(h), nm_u8, 0, n, layer.norm_y2.weight, layer.norm_y2.bias, seed, salt + 5, p, False)
        E, Eb = ResidualLN.apply(E, nE, nm_u8, 2, n, layer.normE1.weight, layer.normE1.bias, seed, salt + 6, p, bf16)
        h = wide_linear(layer.linE1, Eb if bf16 else E, act=(seed, salt + 7, p))
        E, Eb = ResidualLN.apply(E, wide_linear(layer.linE2, h), nm_u8, 0, n, layer.normE2.weight, layer.normE2.bias, seed, salt + 8, p, bf16)
        return X, E, y, Eb
    if layer.self_attn.improved:
        lanes.to_pair(X, y)
        nX, nE, ny = _node_edge_block_improved(layer.self_attn, X, E, y, nm_u8, xm, em)
        lanes.to_node(nX, ny)
    else:
        nX, nE, ny = _node_edge_block(layer.self_attn, X, E, y, nm_u8, xm, em, E.to(torch.bfloat16) if bf16 else None, lanes)
    with lanes.on_node():
        X = AddLayerNorm.apply(X, drop(nX), layer.normX1.weight, layer.normX1.bias)
        y = AddLayerNorm.apply(y, drop(ny), layer.norm_y1.weight, layer.norm_y1.bias)
        X = AddLayerNorm.apply(X, drop(layer.linX2(drop(F.relu(layer.linX1(X))))), layer.normX2.weight, layer.normX2.bias)
        y = AddLayerNorm.apply(y, drop(layer.lin_y2(drop(F.relu(layer.lin_y1(y))))), layer.norm_y2.weight, layer.norm_y2.bias)
    E = AddLayerNorm.apply(E, drop(nE), layer.normE1.weight, layer.normE1.bias)
    if bf16:       # the edge FFN's 128-wide hidden tensor in bf16 too
        E = AddLayerNorm.apply(E, drop(wide_linear(layer.linE2, drop(F.relu(wide_linear(layer.linE1, E.to(torch.bfloat16))))).float()),
                               layer.normE2.weight, layer.normE2.bias)
    else:
        E = AddLayerNorm.apply(E, drop(wide_linear(layer.linE2, drop(F.relu(wide_linear(layer.linE1, E))))), layer.normE2.weight, layer.normE2.bias)
    return X, E, y, None


def _fusable(layer, X, E, y):
    """the fused blocks are built for row widths 16..256 (powers of two) and hidden sizes that are multiples of 4"""
    ok = X.shape[-1] in FUSED_WIDTHS and E.shape[-1] in FUSED_WIDTHS and y.shape[-1] in FUSED_WIDTHS
    hid = (layer.linX1.out_features, layer.linE1.out_features, layer.lin_y1.out_features)
    return ok and not layer.self_attn.improved and all(h % 4 == 0 for h in hid)


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


# 'fp32' (default: every tensor and GEMM fp32, what the 1e-4 gradient checks run) or 'bf16': mixed precision for the per-pair tensors of
# the layers (the reference trains under fp16 autocast: diffusion_abstract.py:181-183) -- bf16 storage and bf16 GEMM operands with fp32
# accumulation for E1, E2, Z, the edge FFN's hidden tensor and their gradients; fp32 master weights, fp32 LayerNorms / softmax / losses.
# Set per module (`module.train_precision = 'bf16'`) or process-wide here / by DA_TRAIN_PRECISION.
PRECISION = _os.environ.get('DA_TRAIN_PRECISION', 'fp32')


def forward_train(m, X, E, y, node_mask, atom_map_numbers=None, pos_encodings=None, mol_assignments=None, precision=None):
    """Differentiable forward of the denoiser module `m`; returns (X logits, E logits, y) like the module's forward.
    `precision` overrides the module's / the process's training precision (the fp32 inference mode passes 'fp32')."""
    bf16 = precision or getattr(m, 'train_precision', None) or PRECISION
    if bf16 not in ('fp32', 'bf16'):
        raise ValueError(f"train precision {bf16!r}: 'fp32' or 'bf16'")
    bf16 = bf16 == 'bf16'
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
    lanes = _Lanes(X.device, SIDE_STREAM and getattr(m, 'train_side_stream', True))
    fused = FUSED and getattr(m, 'train_fused', True)
    seed = _dropout_seed(X.device) if fused and m.training else None
    lanes.to_node(Xh, yh, xm, seed)
    Eb = None
    for k, layer in enumerate(m.tf_layers):
        fuse = (seed, 16 * k) if fused and _fusable(layer, Xh, Eh, yh) else None
        Xh, Eh, yh, Eb = _layer(layer, Xh, Eh, yh, nm_u8, xm, em, m.training, bf16, lanes, Eb, fuse)
    lanes.to_pair(Xh, yh)
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
