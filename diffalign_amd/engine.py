"""Host-side driver of the denoiser forward: packs the nn.Module's parameters once (re-packs when they change) and
issues the C-ABI calls of include/diffalign_hip.h in order.  PyTorch only owns memory and the stream."""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Optional

import torch

from . import _lib as L
from . import pack

N_STAT_CHUNKS = 32                    # most partial sums per graph of the separate Etoy statistics pass (stat_chunks)
N_XSTAT_CHUNKS = 8
# Etoy statistics of layers >= 1 from the edge kernel's tail (da_edge_layer_stats) instead of a separate pass.  A ticket of the tail is
# one (graph, chunk of pairs): about 1,300 pairs per ticket is where the counter's round trip and the per-ticket fold stop showing
# (64 tickets per graph at n = 300, 16 at n = 140; measured steps at B = 64, n = 140: 4.23 / 4.08 / 4.03 / 4.18 ms with 4 / 8 / 16 / 64).
# Below n = 112 the separate pass wins (B = 100, n = 72: 2.72 ms against 2.78 with 4 tickets and 3.15 with 64), from there on the tail.
# DA_STATS_TAIL=0 / 1 forces the separate pass / the tail at every size, DA_TAIL_CHUNKS fixes the ticket count.
_STATS_TAIL_ENV = os.environ.get('DA_STATS_TAIL')
STATS_TAIL = _STATS_TAIL_ENV != '0'
STATS_TAIL_MIN_N = 0 if _STATS_TAIL_ENV == '1' else 112
_TAIL_CHUNKS_ENV = os.environ.get('DA_TAIL_CHUNKS')


# Node tail of a layer: below this many rows the 16-row-tile kernel (two waves per SIMD; twice the weight stream per row) is the
# faster one: 21.5 against 28.2 us at 7,200 rows, 22.5 / 28.8 at 8,960, 40.3 / 32.5 at 19,200 (tools/bench_node_chain.py)
XPOST16_MAX_ROWS = int(os.environ.get('DA_XPOST16_MAX_ROWS', '12000'))


_STAT_CHUNKS_ENV = os.environ.get('DA_STAT_CHUNKS')


def stat_chunks(n: int) -> int:
    """Workgroups per graph of the separate Etoy statistics pass (da_edge_stats): about 64 KB of the bf16 hidden tensor each.  With the
    32 that suit n >= 130 a workgroup at n = 72 streams 20 KB and spends its time in the fold of its partial sums."""
    if _STAT_CHUNKS_ENV:
        return int(_STAT_CHUNKS_ENV)
    return max(1, min(N_STAT_CHUNKS, (n * n) // 512))


def tail_chunks(n: int) -> int:
    """Statistics tickets per graph of the edge kernel's tail at graph size n."""
    if _TAIL_CHUNKS_ENV:
        return int(_TAIL_CHUNKS_ENV)
    return max(4, min(64, (n * n) // 1300))


def use_stats_tail(n: int) -> bool:
    return STATS_TAIL and n >= STATS_TAIL_MIN_N


# One arrangement of everything else (round 3; the alternates' A/B numbers and bit-equality evidence are in DESIGN.md 3.4): node chain on
# the caller's stream with the y chain and the node-feature work on two side streams (_layers_node_on_main), fused node input stage,
# first layer from the input stage's class table, edge output as mirrored tiles with the reverse step drawn inside.  Under HIP-graph
# capture (no side streams) the same kernels run on one stream (layer_forward).  The two attributes below are test hooks, not switches.
CAPTURE_SIDE_STREAMS = True   # under HIP-graph capture the side streams join the capture through their events (False: one stream)
TABLE_IN = True        # first edge layer reads its input rows from the input stage's table (da_edge_in_classes / da_edge_layer_table)
SPLIT_TAIL = True      # load-balance the edge kernel's last round (da_edge_split_scratch_bytes); False: whole rows only
PROFILE_EVENTS = None   # bench.py sets this to a list: (start event, end event, share of a full launch's algorithmic work) per da_edge_layer launch


def _i(v):
    return C.c_int(int(v))


class _PtrArray:
    """ctypes array of device pointers (host memory) + the tensors it references kept alive."""

    def __init__(self, tensors):
        self.keep = list(tensors)
        self.arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class PackedModel:
    def __init__(self, sd: Dict[str, torch.Tensor], device, n_layers: int, n_head: int, prefix: str = ''):
        g = lambda k: sd[prefix + k]
        dev = device
        f32 = lambda t: t.detach().float().contiguous().to(dev)
        bw = lambda k: pack.bf16_weight(g(k)).to(dev)
        self.n_layers, self.n_head = n_layers, n_head
        self.dx = g('mlp_in_X.2.weight').shape[0]
        self.de = g('mlp_in_E.2.weight').shape[0]
        self.dy = g('mlp_in_y.2.weight').shape[0]
        self.hx = g('mlp_in_X.0.weight').shape[0]
        self.he = g('mlp_in_E.0.weight').shape[0]
        self.hy = g('mlp_in_y.0.weight').shape[0]
        self.dffx = g('tf_layers.0.linX1.weight').shape[0]
        self.dffe = g('tf_layers.0.linE1.weight').shape[0]
        self.dffy = g('tf_layers.0.lin_y1.weight').shape[0]
        self.kx = g('mlp_out_X.2.weight').shape[0]
        self.ke = g('mlp_out_E.2.weight').shape[0]
        self.dx_in = g('mlp_in_X.0.weight').shape[1]
        self.de_in = g('mlp_in_E.0.weight').shape[1]
        self.dy_in = g('mlp_in_y.0.weight').shape[1]
        lib = L.lib()
        if lib.da_edge_blob_bytes(_i(self.de), _i(self.dx), _i(self.dffe)) < 0:
            raise L.DiffAlignHipError(f'hidden dims (de={self.de}, dx={self.dx}, dim_ffE={self.dffe}) are not built into '
                                      'libdiffalign_hip.so; built: (64,256,128) and (32,64,64)')
        self.in_X = [bw('mlp_in_X.0.weight'), f32(g('mlp_in_X.0.bias')), bw('mlp_in_X.2.weight'), f32(g('mlp_in_X.2.bias'))]
        # fused node input stage (da_x_in) where its widths are built
        self.x_in = (pack.x_in_blob(sd, prefix).to(dev)
                     if L.lib().da_x_in_blob_bytes(_i(g('mlp_in_X.0.weight').shape[1]), _i(self.hx), _i(self.dx)) > 0 else None)
        self.in_y = [f32(g('mlp_in_y.0.weight')), f32(g('mlp_in_y.0.bias')), f32(g('mlp_in_y.2.weight')), f32(g('mlp_in_y.2.bias'))]
        self.in_E = pack.edge_in_blob(sd, prefix).to(dev)
        self.out_X = [bw('mlp_out_X.0.weight'), f32(g('mlp_out_X.0.bias')), bw('mlp_out_X.2.weight'), f32(g('mlp_out_X.2.bias'))]
        self.out_E = pack.edge_out_blob(sd, prefix).to(dev)
        self.skip = None
        if prefix + 'skip_scaling' in sd:
            self.skip = (f32(g('skip_scaling')), f32(g('skip_scaling_2')))
        self.fused_x = lib.da_x_post_blob_bytes(_i(self.dx), _i(self.dffx)) > 0     # else: the unfused da_linear chain
        self.fused_qkv = self.dx in (256, 64)                                       # else: da_linear + da_kv_pack
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)   # da_edge_layer_stats' work counter: the kernel leaves it zero
        self.layers = []
        for l in range(n_layers):
            p = f'tf_layers.{l}'
            a = p + '.self_attn'
            Wqkv = torch.cat([g(a + '.q.weight'), g(a + '.k.weight'), g(a + '.v.weight')], 0)
            bqkv = torch.cat([g(a + '.q.bias'), g(a + '.k.bias'), g(a + '.v.bias')], 0)
            tr = lambda t: t.detach().float().t().contiguous().to(dev)      # transposed [K][O] for block_matvec_T
            # every weight of the y chain is stored k-major ((in, out): consecutive threads read consecutive outputs)
            film = [(tr if k.endswith('weight') else f32)(g(a + k)) for k in ('.y_e_add.weight', '.y_e_add.bias', '.y_e_mul.weight', '.y_e_mul.bias',
                                                                              '.y_x_add.weight', '.y_x_add.bias', '.y_x_mul.weight', '.y_x_mul.bias')]
            film += [tr(g(a + '.e_out.weight')), f32(g(a + '.e_out.bias'))]
            yup = [tr(g(a + '.x_y.lin.weight')), f32(g(a + '.x_y.lin.bias')), tr(g(a + '.e_y.lin.weight')), f32(g(a + '.e_y.lin.bias')),
                   tr(g(a + '.y_out.0.weight')), f32(g(a + '.y_out.0.bias')), tr(g(a + '.y_out.2.weight')), f32(g(a + '.y_out.2.bias')),
                   f32(g(p + '.norm_y1.weight')), f32(g(p + '.norm_y1.bias')), tr(g(p + '.lin_y1.weight')), f32(g(p + '.lin_y1.bias')),
                   tr(g(p + '.lin_y2.weight')), f32(g(p + '.lin_y2.bias')), f32(g(p + '.norm_y2.weight')), f32(g(p + '.norm_y2.bias'))]
            self.layers.append(dict(
                blob=pack.edge_layer_blob({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, p).to(dev),
                Wqkv=pack.bf16_weight(Wqkv).to(dev), bqkv=f32(bqkv),
                Wxo=bw(a + '.x_out.weight'), bxo=f32(g(a + '.x_out.bias')),
                W1=bw(p + '.linX1.weight'), b1=f32(g(p + '.linX1.bias')), W2=bw(p + '.linX2.weight'), b2=f32(g(p + '.linX2.bias')),
                n1=(f32(g(p + '.normX1.weight')), f32(g(p + '.normX1.bias'))),
                n2=(f32(g(p + '.normX2.weight')), f32(g(p + '.normX2.bias'))),
                xpost=(pack.x_post_blob({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, p).to(dev)
                       if self.fused_x else None),
                xpost16=(pack.x_post16_blob({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, p).to(dev)
                         if self.fused_x and lib.da_x_post16_blob_bytes(_i(self.dx), _i(self.dffx)) > 0 else None),
                film=_PtrArray(film), yup=_PtrArray(yup)))


def _linear(A, lda, W, bias, rowmask, out, ldc, M, N, K, relu):
    L.check(L.lib().da_linear(L.ptr(A) if isinstance(A, torch.Tensor) else A, _i(lda), L.ptr(W), _i(W.shape[1]), L.ptr(bias),
                              L.ptr(rowmask), L.ptr(out) if isinstance(out, torch.Tensor) else out, _i(ldc), _i(M), _i(N),
                              _i(K), _i(1 if relu else 0), L.stream()), 'da_linear')


def edge_stats(E_bf16, B, n, de, out=None):
    st = out if out is not None else torch.empty(B, stat_chunks(n), 4, de, dtype=torch.float32, device=E_bf16.device)
    L.check(L.lib().da_edge_stats(L.ptr(E_bf16), L.ptr(st), _i(B), _i(n), _i(de), _i(st.shape[1]), L.stream()), 'da_edge_stats')
    return st


def film_buffers(pm: PackedModel, B, dev):
    return (torch.empty(B, pm.dx, device=dev), torch.empty(B, pm.de, device=dev), torch.empty(B, pm.dx, device=dev),
            torch.empty(B, pm.dx, device=dev))


def layer_film(pm: PackedModel, lw: dict, y, B, out):
    """FiLM vectors of a layer from y (layers.py:241-262): y_e_mul + 1, the per-graph e_out bias, y_x_add, y_x_mul + 1."""
    ye2p1, ebias, yx1, yx2p1 = out
    L.check(L.lib().da_y_film(L.ptr(y), lw['film'].arr, L.ptr(ye2p1), L.ptr(ebias), L.ptr(yx1), L.ptr(yx2p1), _i(B), _i(pm.dy),
                              _i(pm.dx), _i(pm.de), L.stream()), 'da_y_film')
    return out


def edge_stats_after_in(scratch, E_bf16, B, n, de_in, de, out=None):
    """Statistics partials of the input stage's output: from the class counts da_edge_in_sym left in `scratch` for the graphs
    its table served (no pass over the hidden tensor), da_edge_stats for the others."""
    nch = stat_chunks(n)
    st = out if out is not None else torch.empty(B, nch, 4, de, dtype=torch.float32, device=E_bf16.device)
    lib = L.lib()
    if de_in <= 8:
        L.check(lib.da_edge_stats_table(L.ptr(scratch), L.ptr(st), _i(B), _i(de_in), _i(de), _i(nch), L.stream()),
                'da_edge_stats_table')
    L.check(lib.da_edge_stats_skip(L.ptr(E_bf16), L.ptr(st), L.ptr(scratch), _i(B), _i(n), _i(de), _i(nch), L.stream()),
            'da_edge_stats_skip')
    return st


def _qkv(pm: PackedModel, lw: dict, X, nm_u8, B, n, out=None):
    """q, k, v projections of a layer: Q (R, ldq) fp32 and K, V as the edge kernel's bf16 fragments.  `out`: preallocated (Qm, kvf)."""
    dev, dx, R, lib, S = X.device, pm.dx, B * n, L.lib(), L.stream()
    Qm, kvf = out if out is not None else _qkv_buffers(pm, B, n, dev)
    if pm.fused_qkv:
        L.check(lib.da_qkv_frag(L.ptr(X), L.ptr(lw['Wqkv']), L.ptr(lw['bqkv']), L.ptr(nm_u8), L.ptr(Qm), L.ptr(kvf), _i(B), _i(n),
                                _i(dx), S), 'da_qkv_frag')
    else:
        _linear(X, dx, lw['Wqkv'], lw['bqkv'], nm_u8, Qm, 3 * dx, R, 3 * dx, dx, False)
        fsz = 4
        L.check(lib.da_kv_pack(C.c_void_p(Qm.data_ptr() + dx * fsz), C.c_void_p(Qm.data_ptr() + 2 * dx * fsz), _i(3 * dx),
                               L.ptr(kvf), _i(B), _i(n), _i(dx), S), 'da_kv_pack')
    return Qm, kvf


def _qkv_buffers(pm: PackedModel, B, n, dev):
    kvf = torch.empty(L.lib().da_kv_frag_bytes(_i(B), _i(n), _i(pm.dx)), dtype=torch.uint8, device=dev)
    return torch.empty(B * n, pm.dx if pm.fused_qkv else 3 * pm.dx, device=dev), kvf


def _edge(pm: PackedModel, lw: dict, E_bf16, Qm, kvf, nm_u8, film, B, n, stats_out=None, table=None, rows=None, row_tiles=None, share=None):
    """The fused edge layer (+ the merge of its split rows): E' (bf16) and the attention output newX_in (R, dx) before x_out.
    `table` = (cls_map, scratch) of da_edge_in_classes: the first layer reads its input rows from the input stage's table.
    `rows`: int32 tensor of the query rows b n + i to compute, ascending (da_edge_layer_ex): the others keep what a per-size buffer holds in newX_in (zeros or earlier outputs: finite) and
    nothing in E' (the last layer of a fused reverse step: only the rows of nodes that are free to diffuse are read again).
    `row_tiles`: (B n,) int32, bit jt of a row's word set = E'[row][keys 32 jt ..] is wanted (graph.LastLayersPlan); the other tiles
    compute their attention part only.  `share`: the launch's share of a full launch's algorithmic FLOPs, for PROFILE_EVENTS."""
    dev, dx, de, lib, S = E_bf16.device, pm.dx, pm.de, L.lib(), L.stream()
    ye2p1, ebias, yx1, yx2p1 = film
    E_out = torch.empty_like(E_bf16)
    rows_dev = None
    if isinstance(rows, tuple):                      # (list buffer, its length on the device, an estimate of it): graphs.GraphedChain
        rows, rows_dev, n_rows = rows
        n_rows = max(1, min(int(n_rows), B * n))
    else:
        n_rows = int(rows.numel()) if rows is not None else B * n
    nsplit = lib.da_edge_split_scratch_bytes_rows(C.c_longlong(-n_rows if rows_dev is not None else n_rows), _i(n), _i(dx)) if SPLIT_TAIL else 0
    split = torch.empty(nsplit, dtype=torch.uint8, device=dev) if nsplit > 0 else None
    if rows is None:
        newX_in = torch.empty(B * n, dx, device=dev)
    else:
        # the rows left out keep what the buffer holds, and the node tail reads every row: one buffer per size that starts as zeros and
        # only ever receives attention outputs (finite), instead of a fill launch in front of the step's last edge kernel.  Calls on
        # one stream are ordered; two streams driving the same packed model at once would share it.
        cache = pm.__dict__.setdefault('_rowlist_newx', {})
        newX_in = cache.get((B * n, dx))
        if newX_in is None or newX_in.device != dev:
            newX_in = cache[(B * n, dx)] = torch.zeros(B * n, dx, device=dev)
    ticket = pm.ticket if stats_out is not None else None
    profile = PROFILE_EVENTS is not None and not torch.cuda.is_current_stream_capturing()
    if profile:
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True); ev0.record()
    args = (L.ptr(E_bf16), L.ptr(E_out), L.ptr(Qm), _i(Qm.shape[1]), L.ptr(kvf), L.ptr(nm_u8), L.ptr(ye2p1),
            L.ptr(ebias), L.ptr(yx1), L.ptr(yx2p1), L.ptr(newX_in), L.ptr(lw['blob']),
            C.c_float(1.0 / math.sqrt(dx // pm.n_head)), _i(B), _i(n), _i(de), _i(dx), _i(pm.dffe),
            L.ptr(split), C.c_longlong(split.numel() if split is not None else 0))
    if rows is not None or row_tiles is not None:
        assert table is None and (rows is None or (rows.dtype == torch.int32 and rows.is_contiguous()))
        assert row_tiles is None or (row_tiles.dtype == torch.int32 and row_tiles.is_contiguous() and row_tiles.numel() == B * n)
        N_ = L.ptr(None)
        L.check(lib.da_edge_layer_ex(*args, L.ptr(stats_out), _i(stats_out.shape[1] if stats_out is not None else 0), L.ptr(ticket),
                                     N_, N_, _i(0), L.ptr(rows), _i(n_rows if rows is not None else 0), L.ptr(rows_dev), L.ptr(row_tiles), S),
                'da_edge_layer_ex')
    elif table is not None:
        assert stats_out is None
        L.check(lib.da_edge_layer_table(*args, L.ptr(table[0]), L.ptr(table[1]), _i(pm.de_in), S), 'da_edge_layer_table')
    elif stats_out is not None:
        L.check(lib.da_edge_layer_stats(*args, L.ptr(stats_out), _i(stats_out.shape[1]), L.ptr(ticket), S), 'da_edge_layer_stats')
    else:
        L.check(lib.da_edge_layer(*args, S), 'da_edge_layer')
    if profile:
        ev1.record(); PROFILE_EVENTS.append((ev0, ev1, float(share) if share is not None else n_rows / float(B * n)))
    return E_out, newX_in, split


def _x_post(pm: PackedModel, lw: dict, newX_in, X, nm_u8, B, n, out=None):
    """Node tail of a layer (layers.py:266-273, 340-351): x_out + mask, normX1, FFN X, normX2."""
    dev, dx, R, lib, S = X.device, pm.dx, B * n, L.lib(), L.stream()
    X2 = out if out is not None else torch.empty(B, n, dx, device=dev)
    if lw['xpost16'] is not None and R <= XPOST16_MAX_ROWS:
        L.check(lib.da_x_post16(L.ptr(newX_in), L.ptr(X), L.ptr(nm_u8), L.ptr(lw['xpost16']), L.ptr(X2), _i(R), _i(dx), _i(pm.dffx), S),
                'da_x_post16')
    elif lw['xpost'] is not None:
        L.check(lib.da_x_post(L.ptr(newX_in), L.ptr(X), L.ptr(nm_u8), L.ptr(lw['xpost']), L.ptr(X2), _i(R), _i(dx), _i(pm.dffx), S),
                'da_x_post')
    else:
        newX = torch.empty(R, dx, device=dev)
        _linear(newX_in, dx, lw['Wxo'], lw['bxo'], nm_u8, newX, dx, R, dx, dx, False)
        X1 = torch.empty(R, dx, device=dev)
        L.check(lib.da_add_layernorm(L.ptr(X), L.ptr(newX), L.ptr(lw['n1'][0]), L.ptr(lw['n1'][1]), L.ptr(X1), _i(R), _i(dx), S),
                'da_add_layernorm')
        hbuf = torch.empty(R, pm.dffx, device=dev)
        _linear(X1, dx, lw['W1'], lw['b1'], None, hbuf, pm.dffx, R, pm.dffx, dx, True)
        ff = newX
        _linear(hbuf, pm.dffx, lw['W2'], lw['b2'], None, ff, dx, R, dx, pm.dffx, False)
        L.check(lib.da_add_layernorm(L.ptr(X1), L.ptr(ff), L.ptr(lw['n2'][0]), L.ptr(lw['n2'][1]), L.ptr(X2), _i(R), _i(dx), S),
                'da_add_layernorm')
    return X2


def layer_main(pm: PackedModel, lw: dict, X, E_bf16, y, nm_u8, B, n, film=None, before_edge=None, after_edge=None, stats_out=None):
    """X and E branches of one XEyTransformerLayer (neuralnet/layers.py:204-273, 330-351): FiLM vectors from y, q/k/v, the
    fused edge layer, the fused node tail.  X (B,n,dx) fp32, E (B,n,n,de) bf16, y (B,dy) fp32 -> X', E'.
    `film`: the FiLM vectors if the caller computed them already (layer_film); `before_edge` / `after_edge`: hooks called
    right before / after the edge kernel is enqueued (stream joins of the pipelined y branch); `stats_out`: buffer
    (B, tail_chunks(n), 4, de) that receives the Etoy statistics partials of E_bf16 from the edge kernel's tail."""
    film = film if film is not None else layer_film(pm, lw, y, B, film_buffers(pm, B, X.device))
    Qm, kvf = _qkv(pm, lw, X, nm_u8, B, n)
    if before_edge is not None:
        before_edge()
    E_out, newX_in, _ = _edge(pm, lw, E_bf16, Qm, kvf, nm_u8, film, B, n, stats_out)
    if after_edge is not None:
        after_edge()
    return _x_post(pm, lw, newX_in, X, nm_u8, B, n), E_out


def node_stats(pm: PackedModel, X, B, n, out=None):
    xst = out if out is not None else torch.empty(B, N_XSTAT_CHUNKS, 4, pm.dx, device=X.device)
    L.check(L.lib().da_x_stats(L.ptr(X), L.ptr(xst), _i(B), _i(n), _i(pm.dx), _i(N_XSTAT_CHUNKS), L.stream()), 'da_x_stats')
    return xst


def y_update(pm: PackedModel, lw: dict, y, xst, estats, B, n, out):
    L.check(L.lib().da_y_update(L.ptr(y), L.ptr(xst), L.ptr(estats), lw['yup'].arr, L.ptr(out), _i(B), _i(n), _i(pm.dx),
                                _i(pm.de), _i(pm.dy), _i(pm.dffy), _i(N_XSTAT_CHUNKS), _i(estats.shape[1]), L.stream()), 'da_y_update')
    return out


def y_pre(pm: PackedModel, lw: dict, xst, B, n, out):
    """x_y.lin(Xtoy statistics): the part of y_update that needs only the layer's input X (C-ABI da_y_pre)."""
    L.check(L.lib().da_y_pre(L.ptr(xst), lw['yup'].arr, L.ptr(out), _i(B), _i(n), _i(pm.dx), _i(pm.dy), _i(N_XSTAT_CHUNKS), L.stream()), 'da_y_pre')
    return out


def y_update_film(pm: PackedModel, lw: dict, nxt: Optional[dict], y, xy, estats, B, n, out, film_out):
    """y_update of layer `lw` from the precomputed Xtoy term, with the FiLM vectors of the next layer in the same launch."""
    f = film_out if nxt is not None else (None, None, None, None)
    L.check(L.lib().da_y_update_film(L.ptr(y), L.ptr(xy), L.ptr(estats), lw['yup'].arr, L.ptr(out), nxt['film'].arr if nxt is not None else None,
                                     L.ptr(f[0]), L.ptr(f[1]), L.ptr(f[2]), L.ptr(f[3]), _i(B), _i(n), _i(pm.dx), _i(pm.de), _i(pm.dy),
                                     _i(pm.dffy), _i(estats.shape[1]), L.stream()), 'da_y_update_film')
    return out


def layer_y(pm: PackedModel, lw: dict, X, y, estats, B, n):
    """y branch of one XEyTransformerLayer (layers.py:277-281, 353-359): pooled statistics of the layer's INPUT X and E,
    y_out, the two LayerNorms and the y FFN.  Independent of the X / E branches of the same layer."""
    return y_update(pm, lw, y, node_stats(pm, X, B, n), estats, B, n, torch.empty(B, pm.dy, device=X.device))


def tail_stats_buffer(pm: PackedModel, B, n, dev):
    return torch.empty(B, tail_chunks(n), 4, pm.de, dtype=torch.float32, device=dev)


def layer_forward(pm: PackedModel, lw: dict, X, E_bf16, y, nm_u8, estats, B, n):
    """One XEyTransformerLayer (neuralnet/layers.py:330-361) on the current stream.  estats: Etoy statistics partials of
    E_bf16, or None: the edge kernel's tail computes them (STATS_TAIL) or a separate pass does."""
    tail = tail_stats_buffer(pm, B, n, X.device) if estats is None and use_stats_tail(n) else None
    if estats is None and tail is None:
        estats = edge_stats(E_bf16, B, n, pm.de)
    X2, E_out = layer_main(pm, lw, X, E_bf16, y, nm_u8, B, n, stats_out=tail)
    return X2, E_out, layer_y(pm, lw, X, y, estats if estats is not None else tail, B, n)


_SIDE = {}


def _side_stream(dev):
    """Second HIP stream for the latency-bound tail of the y branch (none under HIP-graph capture).  The pooled statistics
    of a layer's input stay on the main stream (one HBM-bound pass; the edge kernel owns every CU's registers, nothing
    overlaps with it; on a third stream, beside the node tail and the next q/k/v, it made the step 1-2 % slower).  What moves is y_update(l) -> y_film(l+1): 64 workgroups and ~15 barriers each, enqueued behind
    the edge kernel of layer l so that they run beside the node tail of layer l and the q/k/v and statistics passes of
    layer l+1; the main stream joins right before the edge kernel of layer l+1.  A second side stream carries what depends on the
    node features only: the Xtoy statistics beside q/k/v, the node input / output paths beside the edge input / output
    stages, and (helpers.posterior_sample) the node half of the reverse step.  Returns (y stream, node stream) or None."""
    if torch.cuda.is_current_stream_capturing() and not CAPTURE_SIDE_STREAMS:
        return None
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SIDE[key]


def _last_plan(last_rows):
    """reverse_step's last_rows (graph.free_rows' tensor, (list, length on the device, estimate), or a graph.LastLayersPlan) as a plan."""
    from .graph import LastLayersPlan
    if last_rows is None or isinstance(last_rows, LastLayersPlan):
        return last_rows
    if isinstance(last_rows, tuple):
        return LastLayersPlan(last_rows[0], last_rows[1], last_rows[2])
    return LastLayersPlan(last_rows)


def _layers_node_on_main(pm: PackedModel, Xh, Eh, yh, nm, st0, film, y_done0, B, n, main, side, side2, pre, table0=None, last_rows=None):
    """The layer stack with the NODE chain on the caller's stream (merge, node tail of layer l, q/k/v of layer l+1 directly behind the
    edge kernel and in front of the next one: no cross-stream event on it) and the y chain on the side stream (edge kernel -> event ->
    y_update(l) + FiLM(l+1) [+ the separate statistics pass of small graphs] -> event the next edge kernel waits on).  Since the y
    chain's kernel takes 25 us instead of 46 (k-major weights, wave-level LayerNorms) it ends before the node chain does, so its two
    events are off the critical path, provided its 1,024-thread workgroups find free CUs: da_x_post runs four row tiles per
    workgroup (150 CUs at 19,200 rows) instead of 600 single-wave workgroups that take a whole SIMD's registers each on every CU.
    Gap between two edge kernels 134 -> 115 us; steps in alternating same-box runs: 13.16 -> 13.01 ms (headline), 4.02 -> 3.91 ms
    (B = 64, n = 140), 2.70 -> 2.65 ms (B = 100, n = 72), 7.26 -> 7.11 ms (USPTO-like).
    `pre` = (Qm, kvf, xst, xy) of layer 0 from the input stage (model_forward)."""
    dev = Xh.device
    Qm, kvf, xst, xy = pre
    plan = _last_plan(last_rows)
    if y_done0 is not None:
        main.wait_event(y_done0)                     # FiLM vectors of layer 0 (y stream of the input stage)
    xs_done = None                                   # Xtoy term of the coming layer complete (third stream)
    st_pass = None                                   # statistics of this layer's input from a separate pass (graphs too small for the tail)
    keep = []
    for li, lw in enumerate(pm.layers):
        nxt = pm.layers[li + 1] if li + 1 < len(pm.layers) else None
        # the last layer of a fused reverse step with a row list: its y output is read by nobody (the step returns z_s), so neither the
        # Etoy statistics of its input (the edge kernel's tail) nor its y update are computed; the stream joins stay as they are
        no_y = nxt is None and li > 0 and plan is not None
        # the layer before it: its E' is the last layer's input, which reads the rows of the list only -- the other rows compute their
        # attention part (their node features are every row's keys in the last layer) and form no E'; nobody reads that E's statistics
        pre_last = plan is not None and li > 0 and li + 2 == len(pm.layers)
        tail = tail_stats_buffer(pm, B, n, dev) if li > 0 and use_stats_tail(n) and not no_y else None
        st = st0 if li == 0 else (tail if tail is not None else st_pass)
        st_next = (torch.empty(B, stat_chunks(n), 4, pm.de, dtype=torch.float32, device=dev)
                   if nxt is not None and not use_stats_tail(n) else None)
        X_next = torch.empty(B, n, pm.dx, device=dev)
        y_next = torch.empty(B, pm.dy, device=dev)
        film_next = film_buffers(pm, B, dev) if nxt is not None else None
        qkv_next = _qkv_buffers(pm, B, n, dev) if nxt is not None else None
        xst_next = torch.empty(B, N_XSTAT_CHUNKS, 4, pm.dx, device=dev) if nxt is not None else None
        xy_next = torch.empty(B, pm.dy, device=dev) if nxt is not None else None
        if no_y:
            rows_arg = (plan.rows, plan.n_rows_dev, plan.hint) if plan.n_rows_dev is not None else plan.rows
            E_next, newX_in, split = _edge(pm, lw, Eh, Qm, kvf, nm, film, B, n, tail, rows=rows_arg, row_tiles=plan.tiles_last,
                                           share=plan.share_last if plan.tiles_last is not None else None)
        elif pre_last and plan.tiles_prev is not None:
            E_next, newX_in, split = _edge(pm, lw, Eh, Qm, kvf, nm, film, B, n, tail, row_tiles=plan.tiles_prev,
                                           share=plan.share_prev if plan.share_prev is not None else 1.0)
        else:
            E_next, newX_in, split = _edge(pm, lw, Eh, Qm, kvf, nm, film, B, n, tail, table=table0 if li == 0 else None)
        ev_edge = main.record_event()
        with torch.cuda.stream(side):                # y chain
            side.wait_event(ev_edge)
            if xs_done is not None:
                side.wait_event(xs_done)             # xy of this layer (recorded one edge kernel ago)
            if not no_y:
                y_update_film(pm, lw, nxt, yh, xy, st, B, n, y_next, film_next)
            if st_next is not None and not pre_last:
                edge_stats(E_next, B, n, pm.de, out=st_next)
            ev_y = side.record_event()
        _x_post(pm, lw, newX_in, Xh, nm, B, n, out=X_next)
        if nxt is not None:
            ev_xp = main.record_event()
            with torch.cuda.stream(side2):           # next layer's Xtoy statistics and their x_y.lin term beside its q/k/v
                side2.wait_event(ev_xp)
                node_stats(pm, X_next, B, n, out=xst_next)
                y_pre(pm, nxt, xst_next, B, n, xy_next)
                xs_done = side2.record_event()
            _qkv(pm, nxt, X_next, nm, B, n, out=qkv_next)
        main.wait_event(ev_y)                        # FiLM vectors of the next layer / the stack's y output
        keep.append((Xh, Eh, yh, film, Qm, kvf, xst, xy, st, newX_in, split))
        if len(keep) > 2:
            keep.pop(0)                              # layer l-2: every reader of its buffers has been joined since
        Xh, Eh, yh, film, xst, xy, st_pass = X_next, E_next, (None if no_y else y_next), film_next, xst_next, xy_next, st_next
        if nxt is not None:
            Qm, kvf = qkv_next
    if xs_done is not None:
        main.wait_event(xs_done)
    del keep
    return Xh, Eh, yh


def model_forward(pm: PackedModel, X, E, y, node_mask, atom_map_numbers=None, pos_encodings=None, mol_assignments=None,
                  use_skip=False, step: Optional[dict] = None, edge_skip_uniform=False, align_src=None):
    """GraphTransformerWithY[AtomMapPosEmb].forward (neuralnet/transformer_model_with_y.py:74-111, 620-713).
    `step`: the reverse step fused behind the output stage (helpers.ReverseStep.ctx: z_t, transition tables, noise, masks): the edge
    half is drawn inside the output kernel (da_edge_out_sample) and the edge logits are not materialised (Eo is None); the node half runs behind the node output path.  The drawn state lands in step['zs_X'], step['zs_E'].
    `align_src` (B,n) int32: the skip connection's product index per node when it is not da_align_index's (alignment_type='old').
    `edge_skip_uniform`: input_alignment=True scales the whole edge skip term by skip_scaling_2 (transformer_model_with_y.py:702-704)."""
    dev = X.device
    B, n = X.shape[:2]
    R = B * n
    lib = L.lib()
    S = L.stream()
    X = X.contiguous().float(); E = E.contiguous().float(); y = y.contiguous().float()
    assert X.shape[-1] == pm.dx_in and E.shape[-1] == pm.de_in and y.shape[-1] == pm.dy_in, 'input feature widths'
    nm = L.as_u8(node_mask)
    main, sides = torch.cuda.current_stream(), _side_stream(dev)
    side, side2 = sides if sides is not None else (None, None)
    # X / y columns still being written on another stream (compute_extra_data: the eigensolver's features): (event, tensors to hold until
    # this stream has waited).  The node and y input paths wait for it on their streams; this stream enqueues the edge input stage,
    # which reads only E, first.
    pending = step.pop('inputs_ready', None) if step is not None else None
    if pending is not None and (side is None or side2 is None):
        main.wait_event(pending[0]); pending = None
    # ---- input stage.  The node path (two linears, positional encodings, the alignment index of the skip connection) depends
    # on the inputs only: with side streams it runs beside the edge input stage.
    h0 = torch.empty(R, pm.hx, device=dev)
    Xh = torch.empty(B, n, pm.dx, device=dev)
    pos = pos_encodings.contiguous().float() if pos_encodings is not None else None
    src = None
    if use_skip:
        assert pm.skip is not None and atom_map_numbers is not None and mol_assignments is not None
        src = align_src.contiguous() if align_src is not None else torch.empty(B, n, dtype=torch.int32, device=dev)
        am64, mol64 = atom_map_numbers.contiguous(), mol_assignments.contiguous()

    # with the three-stream layer stack, what layer 0 needs from the node features alone (q/k/v, the Xtoy statistics and their x_y.lin
    # term) follows them on their stream, beside the edge input stage; the alignment index, which only the output stage reads, goes last
    early0 = side is not None and side2 is not None
    pre = (*_qkv_buffers(pm, B, n, dev), torch.empty(B, N_XSTAT_CHUNKS, 4, pm.dx, device=dev), torch.empty(B, pm.dy, device=dev)) if early0 else None

    def node_features():
        if pm.x_in is not None:
            L.check(lib.da_x_in(L.ptr(X), _i(pm.dx_in), L.ptr(nm), L.ptr(pos), L.ptr(pm.x_in), L.ptr(Xh), _i(R), _i(pm.hx), _i(pm.dx),
                                L.stream()), 'da_x_in')
        else:
            _linear(X, pm.dx_in, pm.in_X[0], pm.in_X[1], None, h0, pm.hx, R, pm.hx, pm.dx_in, True)
            _linear(h0, pm.hx, pm.in_X[2], pm.in_X[3], None, Xh, pm.dx, R, pm.dx, pm.hx, True)
            L.check(lib.da_x_in_finish(L.ptr(Xh), L.ptr(nm), L.ptr(pos), _i(R), _i(pm.dx), L.stream()), 'da_x_in_finish')
        if pre is not None:
            node_stats(pm, Xh, B, n, out=pre[2])
            y_pre(pm, pm.layers[0], pre[2], B, n, pre[3])
            _qkv(pm, pm.layers[0], Xh, nm, B, n, out=(pre[0], pre[1]))

    def alignment():
        if use_skip and align_src is None:
            L.check(lib.da_align_index(L.ptr(am64, torch.int64), L.ptr(mol64, torch.int64), L.ptr(src), _i(B), _i(n), L.stream()),
                    'da_align_index')

    xin_done = align_done = None
    # Which of the two input chains runs on the caller's stream: a cross-stream event costs ~14 us before the other stream starts, and
    # the node input kernel (150 workgroups) must get its CUs BEFORE the class pass of the edge stage floods the dispatcher with 2,880
    # small ones (side by side from the start it takes 93 instead of 32 us, and the node chain is the longer one).  So the node chain
    # starts at once on the caller's stream and the edge input stage follows on the node side stream (DA_EDGE_IN_ON_SIDE=0: the
    # other way round, as before the stage's table was kept between calls).
    # (Not under HIP-graph capture: replaying a step captured this way faulted inside the runtime, like every other change of what the
    # side streams do at the start of a captured step; captured steps keep the arrangement they had.)
    edge_in_on_side = (side is not None and side2 is not None and not torch.cuda.is_current_stream_capturing()
                       and os.environ.get('DA_EDGE_IN_ON_SIDE', '1') != '0')
    if side2 is None:
        node_features()
        alignment()
    else:
        in_ready = main.record_event()
        if not edge_in_on_side:
            with torch.cuda.stream(side2):
                side2.wait_event(in_ready)
                if pending is not None:
                    side2.wait_event(pending[0])
                node_features()
                xin_done = side2.record_event()
                alignment()
                align_done = side2.record_event()
    yh = torch.empty(B, pm.dy, device=dev)
    film0 = film_buffers(pm, B, dev) if side is not None else None
    y_done0 = None

    def y_input_path():
        L.check(lib.da_y_in(L.ptr(y), L.ptr(pm.in_y[0]), L.ptr(pm.in_y[1]), L.ptr(pm.in_y[2]), L.ptr(pm.in_y[3]), L.ptr(yh), _i(B),
                            _i(pm.dy_in), _i(pm.hy), _i(pm.dy), L.stream()), 'da_y_in')

    if side is None:
        y_input_path()
    else:                                            # y input MLP and the first layer's FiLM vectors beside the edge input stage; enqueued
                                                     # BEFORE that stage (behind it in program order the two kernels also ran behind it)
        with torch.cuda.stream(side):
            side.wait_event(in_ready)
            if pending is not None:
                side.wait_event(pending[0])
            y_input_path()
            layer_film(pm, pm.layers[0], yh, B, film0)
            y_done0 = side.record_event()
    Eh = torch.empty(B, n, n, pm.de, dtype=torch.bfloat16, device=dev)
    # the input stage's scratch (flags, table, class counts) is kept per batch size: its table depends on the weights alone and is
    # evaluated by the first call only (da_edge_in_classes_keep).  Calls on one stream are ordered; a re-packed model is a new object.
    in_keep = pm.__dict__.setdefault('_in_scratch', {})
    sym_flags = in_keep.get((B, dev))
    table_valid = sym_flags is not None and os.environ.get('DA_KEEP_IN_TABLE', '1') != '0'      # (read per call: an A/B switch)
    if sym_flags is None:
        sym_flags = torch.empty(int(lib.da_edge_in_scratch_bytes(_i(B), _i(pm.de_in), _i(pm.de))), dtype=torch.uint8, device=dev)
    # one-hot symmetric inputs (every diffusion state): the first layer takes its input rows from the input stage's table through a
    # one-byte class per pair, and the hidden tensor is neither written here nor read there
    use_table = TABLE_IN and pm.de_in <= 8 and side is not None and side2 is not None
    cls_map = torch.empty(B, n, n, dtype=torch.uint8, device=dev) if use_table else None
    st0 = torch.empty(B, stat_chunks(n), 4, pm.de, dtype=torch.float32, device=dev)

    def edge_input_stage():                          # (buffers allocated above, on the caller's stream, whichever stream runs this)
        if use_table:
            L.check(lib.da_edge_in_classes_keep(L.ptr(E), _i(pm.de_in), L.ptr(nm), L.ptr(pm.in_E), L.ptr(Eh), L.ptr(cls_map), L.ptr(sym_flags),
                                                _i(B), _i(n), _i(pm.de), _i(pm.he), _i(1 if table_valid else 0), L.stream()),
                    'da_edge_in_classes_keep')
            if not torch.cuda.is_current_stream_capturing():     # (a captured call has not run yet: its table does not exist)
                in_keep[(B, dev)] = sym_flags
        else:
            L.check(lib.da_edge_in_sym(L.ptr(E), _i(pm.de_in), L.ptr(nm), L.ptr(pm.in_E), L.ptr(Eh), L.ptr(sym_flags), _i(B), _i(n), _i(pm.de),
                                       _i(pm.he), L.stream()), 'da_edge_in_sym')
        edge_stats_after_in(sym_flags, Eh, B, n, pm.de_in, pm.de, out=st0)

    table0 = (cls_map, sym_flags) if use_table else None
    if edge_in_on_side:
        with torch.cuda.stream(side2):
            side2.wait_event(in_ready)
            edge_input_stage()
            ein_done = side2.record_event()
            alignment()                              # (read by the output stage only)
            align_done = side2.record_event()
        if pending is not None:
            main.wait_event(pending[0])              # (its tensors were allocated on this stream: held until here)
            pending = None
        node_features()
        main.wait_event(ein_done)
    else:
        edge_input_stage()
        if pending is not None:
            main.wait_event(pending[0])              # (its tensors were allocated on this stream: held until here)
            pending = None
        if xin_done is not None:
            main.wait_event(xin_done)
    # ---- layers
    if side is None:
        for li, lw in enumerate(pm.layers):
            Xh, Eh, yh = layer_forward(pm, lw, Xh, Eh, yh, nm, st0 if li == 0 else None, B, n)
    else:
        Xh, Eh, yh = _layers_node_on_main(pm, Xh, Eh, yh, nm, st0, film0, y_done0, B, n, main, side, side2, pre, table0=table0,
                                          last_rows=step.get('last_rows') if step is not None else None)
    # ---- output stage (the node output path beside the edge output stage)
    h1 = torch.empty(R, pm.hx, device=dev)
    lin = torch.empty(R, pm.kx, device=dev)
    Xo = torch.empty(B, n, pm.kx, device=dev)
    s1, s2 = pm.skip if use_skip else (None, None)
    es1 = s2 if edge_skip_uniform else s1            # scale of the mapped pairs' edge skip term

    def node_output_path():
        _linear(Xh, pm.dx, pm.out_X[0], pm.out_X[1], None, h1, pm.hx, R, pm.hx, pm.dx, True)
        _linear(h1, pm.hx, pm.out_X[2], pm.out_X[3], None, lin, pm.kx, R, pm.kx, pm.hx, False)
        L.check(lib.da_x_out_finish(L.ptr(lin), _i(pm.kx), L.ptr(X), _i(pm.dx_in), L.ptr(src), L.ptr(s1), L.ptr(Xo), _i(B), _i(n),
                                    _i(pm.kx), L.stream()), 'da_x_out_finish')
        if step is not None:                         # node half of the reverse step (posterior + draw + apply_mask), latency-bound
            c = step
            N_ = L.ptr(None)
            L.check(lib.da_posterior_sample(L.ptr(Xo), N_, L.ptr(c['zt_X']), N_, L.ptr(nm), L.ptr(c['Qt_x']), L.ptr(c['Qsb_x']), L.ptr(c['Qtb_x']),
                                            N_, N_, N_, C.c_float(c['temp'][0]), C.c_float(c['temp'][1]), L.ptr(c['noise_X']), N_,
                                            L.ptr(c['orig_X']), N_, L.ptr(c['mx']), N_, N_, N_, L.ptr(c['zs_X']), N_, _i(B), _i(n), _i(pm.kx),
                                            _i(pm.ke), L.stream()), 'da_posterior_sample')

    xout_done = None
    if side2 is None:
        node_output_path()
    else:
        xh_ready = main.record_event()
        with torch.cuda.stream(side2):
            side2.wait_event(xh_ready)
            node_output_path()
            xout_done = side2.record_event()
    Eo = torch.empty(B, n, n, pm.ke, device=dev) if step is None else None
    if align_done is not None:
        main.wait_event(align_done)                  # the skip connection's index (node stream of the input stage; long done)
    if step is not None:
        c = step
        keep = 1 if c.get('keep_fixed') and c['me'] is not None and c['orig_E'] is not None else 0
        L.check(lib.da_edge_out_sample_ex(L.ptr(Eh), L.ptr(E), _i(pm.de_in), L.ptr(pm.out_E), L.ptr(src), L.ptr(es1), L.ptr(s2),
                                          L.ptr(c['zt_E']), L.ptr(nm), L.ptr(c['Qt_e']), L.ptr(c['Qsb_e']), L.ptr(c['Qtb_e']),
                                          C.c_float(c['temp'][1]), L.ptr(c['noise_E']), L.ptr(c['seeds']),
                                          int(c['call']), L.ptr(c['orig_E']), L.ptr(c['me']), L.ptr(c['zs_E']), _i(B), _i(n), _i(pm.de), _i(pm.he),
                                          _i(pm.ke), _i(keep), S), 'da_edge_out_sample_ex')
    else:
        L.check(lib.da_edge_out_tiles(L.ptr(Eh), L.ptr(E), _i(pm.de_in), L.ptr(pm.out_E), L.ptr(src), L.ptr(es1), L.ptr(s2), L.ptr(Eo),
                                      _i(B), _i(n), _i(pm.de), _i(pm.he), _i(pm.ke), S), 'da_edge_out_tiles')
    if xout_done is not None:
        main.wait_event(xout_done)
    return Xo, Eo, yh
