"""Host-side weight pre-packing into the MFMA fragment order the HIP kernels read (csrc/mfma_util.h).

A fragment is 64 lanes x 8 bf16 (1 KiB), lane-linear.  For lane l (r = l & 31, h = l >> 5) and element j, every
fragment used here has the form   F[l][j] = W[32*tile + r][k(s,h,j)]   with W an nn.Linear weight (out, in) and
  natural k: 16s + 8h + j            (operand read straight from memory)
  C-order k: 16s + 8(j>>2) + 4h + (j&3)   (operand built from accumulator registers)
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

_L = torch.arange(64)
_R = (_L & 31)[:, None]                      # (64,1)
_H = (_L >> 5)[:, None]
_J = torch.arange(8)[None, :]                # (1,8)
_K_NAT = 8 * _H + _J                         # + 16 s
_K_C = 8 * (_J >> 2) + 4 * _H + (_J & 3)     # + 16 s


def _frag(W: torch.Tensor, tile: int, s: int, corder: bool, koff: int = 0) -> torch.Tensor:
    k = (_K_C if corder else _K_NAT) + 16 * s + koff
    return W[32 * tile + _R, k]              # (64,8)


def _pad2(W: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    out = torch.zeros(rows, cols, dtype=torch.float32)
    out[:W.shape[0], :W.shape[1]] = W
    return out


def _blob(frags: List[torch.Tensor], params: List[torch.Tensor]) -> torch.Tensor:
    fb = torch.stack(frags).to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1)
    pb = torch.cat([p.float().reshape(-1) for p in params]).contiguous().view(torch.uint8).reshape(-1)
    return torch.cat([fb, pb]).contiguous()


def bf16_weight(W: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight (N,K) -> bf16 in MFMA B-fragment order for da_linear, shape (Npad32/32 * Kpad16/16 * 64, 8) viewed
    as (Npad32, Kpad16): fragment (nt, ks), lane (r, h), element j = W[32nt + r][16ks + 8h + j]; fragments [nt][ks]."""
    N, K = W.shape
    Np, Kp = (N + 31) // 32 * 32, (K + 15) // 16 * 16
    Wp = _pad2(W.detach().float().cpu(), Np, Kp).to(torch.bfloat16)
    # (nt, r, ks, h, j) -> (nt, ks, h, r, j)
    Wf = Wp.view(Np // 32, 32, Kp // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    return Wf.view(Np, Kp)


def edge_layer_blob(sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """Blob for da_edge_layer; p = 'tf_layers.{l}' key prefix.  Layout = EdgeCfg in csrc/k_edge_layer.hip."""
    g = lambda k: sd[p + k].detach().float().cpu()
    Wm, Wa, Wo = g('.self_attn.e_mul.weight'), g('.self_attn.e_add.weight'), g('.self_attn.e_out.weight')
    F1, F2 = g('.linE1.weight'), g('.linE2.weight')
    dx, de = Wm.shape
    dff = F1.shape[0]
    ncb, ks1, mt_n, ht_n, ksf = dx // 32, de // 16, de // 32, dff // 32, dff // 16
    frags = []
    for cb in range(ncb):
        for W in (Wm, Wa):
            for s in range(ks1):
                frags.append(_frag(W, cb, s, True))
    for cb in range(ncb):
        for mt in range(mt_n):
            for s in range(2):
                frags.append(_frag(Wo, mt, s, True, koff=32 * cb))
    for ht in range(ht_n):
        for s in range(ks1):
            frags.append(_frag(F1, ht, s, True))
    for mt in range(mt_n):
        for s in range(ksf):
            frags.append(_frag(F2, mt, s, True))
    params = [g('.self_attn.e_mul.bias') + 1.0, g('.self_attn.e_add.bias'), g('.normE1.weight'), g('.normE1.bias'),
              g('.linE1.bias'), g('.linE2.bias'), g('.normE2.weight'), g('.normE2.bias')]
    return _blob(frags, params)


def x_post_blob(sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """Blob for da_x_post; p = 'tf_layers.{l}' key prefix.  Layout = XPostCfg in csrc/k_x_post.hip."""
    g = lambda k: sd[p + k].detach().float().cpu()
    Wxo, W1, W2 = g('.self_attn.x_out.weight'), g('.linX1.weight'), g('.linX2.weight')
    dx, dff = Wxo.shape[0], W1.shape[0]
    frags = [_frag(Wxo, mt, s, False) for mt in range(dx // 32) for s in range(dx // 16)]
    frags += [_frag(W1, ht, s, True) for ht in range(dff // 32) for s in range(dx // 16)]
    frags += [_frag(W2, mt, s, True) for mt in range(dx // 32) for s in range(dff // 16)]
    params = [g('.self_attn.x_out.bias'), g('.normX1.weight'), g('.normX1.bias'), g('.linX1.bias'), g('.linX2.bias'),
              g('.normX2.weight'), g('.normX2.bias')]
    return _blob(frags, params)


_R16 = (_L & 15)[:, None]
_G16 = (_L >> 4)[:, None]
_K16_NAT = 8 * _G16 + _J                           # + 32 s
_K16_C = 16 * (_J >> 2) + 4 * _G16 + (_J & 3)      # + 32 s


def _frag16(W: torch.Tensor, tile: int, s: int, corder: bool) -> torch.Tensor:
    """Fragment of v_mfma_f32_16x16x32_bf16 (csrc/k_x_post.hip, 16-row kernel): lane (r = l & 15, g = l >> 4), element e holds
    W[16 tile + r][k], k = 32 s + 8 g + e (natural) or 32 s + 16 (e >> 2) + 4 g + (e & 3) (two accumulator tiles packed)."""
    return W[16 * tile + _R16, (_K16_C if corder else _K16_NAT) + 32 * s]


def x_post16_blob(sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """Blob for da_x_post16; p = 'tf_layers.{l}' key prefix.  Layout = XPost16Cfg in csrc/k_x_post.hip: the output tiles of the three
    GEMMs numbered through, dx / 32 fragments each."""
    g = lambda k: sd[p + k].detach().float().cpu()
    Wxo, W1, W2 = g('.self_attn.x_out.weight'), g('.linX1.weight'), g('.linX2.weight')
    dx, dff = Wxo.shape[0], W1.shape[0]
    assert dx == dff and dx % 32 == 0
    ks = dx // 32
    frags = [_frag16(Wxo, t, s, False) for t in range(dx // 16) for s in range(ks)]
    frags += [_frag16(W1, t, s, True) for t in range(dx // 16) for s in range(ks)]
    frags += [_frag16(W2, t, s, True) for t in range(dx // 16) for s in range(ks)]
    params = [g('.self_attn.x_out.bias'), g('.normX1.weight'), g('.normX1.bias'), g('.linX1.bias'), g('.linX2.bias'),
              g('.normX2.weight'), g('.normX2.bias')]
    return _blob(frags, params)


def x_in_blob(sd, prefix='') -> torch.Tensor:
    """Blob for da_x_in (mlp_in_X).  Layout = XInCfg in csrc/k_x_in.hip: W0 fragments [hidden/32][ceil(dx_in/16)] in natural k order
    (K zero padded), W1 fragments [dx/32][hidden/16] in C order, then b0, b1."""
    g = lambda k: sd[prefix + k].detach().float().cpu()
    W0, W1 = g('mlp_in_X.0.weight'), g('mlp_in_X.2.weight')
    hx, dx_in = W0.shape
    dx = W1.shape[0]
    ks0 = (dx_in + 15) // 16
    W0p = _pad2(W0, hx, 16 * ks0)
    frags = [_frag(W0p, ht, s, False) for ht in range(hx // 32) for s in range(ks0)]
    frags += [_frag(W1, mt, s, True) for mt in range(dx // 32) for s in range(hx // 16)]
    return _blob(frags, [g('mlp_in_X.0.bias'), g('mlp_in_X.2.bias')])


def edge_in_blob(sd, prefix='') -> torch.Tensor:
    g = lambda k: sd[prefix + k].detach().float().cpu()
    W0, W1 = g('mlp_in_E.0.weight'), g('mlp_in_E.2.weight')
    he, de = W0.shape[0], W1.shape[0]
    assert W0.shape[1] <= 16, 'edge input width > 16 is not built'
    W0p = _pad2(W0, he, 16)
    frags = [_frag(W0p, ht, 0, False) for ht in range(he // 32)]
    for mt in range(de // 32):
        for s in range(he // 16):
            frags.append(_frag(W1, mt, s, True))
    return _blob(frags, [g('mlp_in_E.0.bias'), g('mlp_in_E.2.bias')])


def edge_out_blob(sd, prefix='') -> torch.Tensor:
    g = lambda k: sd[prefix + k].detach().float().cpu()
    W0, W1 = g('mlp_out_E.0.weight'), g('mlp_out_E.2.weight')
    he, de = W0.shape
    ke = W1.shape[0]
    assert ke <= 8
    W1p = _pad2(W1, 32, he)
    frags = []
    for ht in range(he // 32):
        for s in range(de // 16):
            frags.append(_frag(W0, ht, s, True))
    for s in range(he // 16):
        frags.append(_frag(W1p, 0, s, True))
    b1 = torch.zeros(32); b1[:ke] = g('mlp_out_E.2.bias')
    return _blob(frags, [g('mlp_out_E.0.bias'), b1])
