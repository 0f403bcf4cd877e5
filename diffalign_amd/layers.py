"""nn.Module shells with the reference's class names, constructor signatures and state-dict keys
(diffalign/neuralnet/layers.py).  They only HOLD parameters; the inference forward runs in the HIP library (engine.py), the training forward in
train_forward.py (which calls the nn.Linear members as the library GEMMs they are)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import Linear, LayerNorm

from . import _lib as L


class Xtoy(nn.Module):
    """layers.py:15-29"""
    def __init__(self, dx, dy):
        super().__init__()
        self.lin = nn.Linear(4 * dx, dy)


class Etoy(nn.Module):
    """layers.py:31-47"""
    def __init__(self, d, dy):
        super().__init__()
        self.lin = nn.Linear(4 * d, dy)


class NodeEdgeBlock(nn.Module):
    """layers.py:166-202 (parameters); forward is fused into XEyTransformerLayer's HIP path.
    `improved`: the parameters of NodeEdgeBlockImproved (layers.py:49-85) are the same set under the same names (its y_e_mul /
    y_e_add exist but are unused); only the arithmetic differs (train_forward._node_edge_block_improved)."""
    def __init__(self, dx, de, dy, n_head, improved=False, **kwargs):
        super().__init__()
        self.improved = improved
        assert dx % n_head == 0, f"dx: {dx} -- nhead: {n_head}"
        self.dx, self.de, self.dy, self.df, self.n_head = dx, de, dy, dx // n_head, n_head
        self.q = Linear(dx, dx); self.k = Linear(dx, dx); self.v = Linear(dx, dx)
        self.e_add = Linear(de, dx); self.e_mul = Linear(de, dx)
        self.y_e_mul = Linear(dy, dx); self.y_e_add = Linear(dy, dx)
        self.y_x_mul = Linear(dy, dx); self.y_x_add = Linear(dy, dx)
        self.y_y = Linear(de, de)        # present in checkpoints, its output is discarded (layers.py:277-280)
        self.x_y = Xtoy(dx, dy); self.e_y = Etoy(de, dy)
        self.x_out = Linear(dx, dx); self.e_out = Linear(dx, de)
        self.y_out = nn.Sequential(nn.Linear(dy, dy), nn.ReLU(), nn.Linear(dy, dy))


class XEyTransformerLayer(nn.Module):
    """layers.py:286-361.  dim_ffy keeps the reference's default 2048 (the model ctor never passes it)."""
    def __init__(self, dx: int, de: int, dy: int, n_head: int, dim_ffX: int = 2048, dim_ffE: int = 128, dim_ffy: int = 2048,
                 dropout: float = 0.1, layer_norm_eps: float = 1e-5, device=None, dtype=None, improved=False) -> None:
        super().__init__()
        assert layer_norm_eps == 1e-5
        kw = {'device': device, 'dtype': dtype}
        self.self_attn = NodeEdgeBlock(dx, de, dy, n_head, improved=improved)                       # layers.py:302
        self.linX1 = Linear(dx, dim_ffX, **kw); self.linX2 = Linear(dim_ffX, dx, **kw)
        self.normX1 = LayerNorm(dx, eps=layer_norm_eps, **kw); self.normX2 = LayerNorm(dx, eps=layer_norm_eps, **kw)
        self.linE1 = Linear(de, dim_ffE, **kw); self.linE2 = Linear(dim_ffE, de, **kw)
        self.normE1 = LayerNorm(de, eps=layer_norm_eps, **kw); self.normE2 = LayerNorm(de, eps=layer_norm_eps, **kw)
        self.lin_y1 = Linear(dy, dim_ffy, **kw); self.lin_y2 = Linear(dim_ffy, dy, **kw)
        self.norm_y1 = LayerNorm(dy, eps=layer_norm_eps, **kw); self.norm_y2 = LayerNorm(dy, eps=layer_norm_eps, **kw)
        self.dropout_p = dropout
