"""Seeded synthetic padded reaction graphs.

Layout per graph: [R reactant slots | SuNo | P product atoms | padding], R = P + nb_dummy
(the reference's graph layout: utils/graph_builder.py:118-192, nb_rct_dummy_nodes=15 from
configs/dataset/uspto-50k.yaml:12).  The *true* reaction built here is "reactants = aligned copy
of the product + dummy 'U' nodes", which is what the reference's own API builds for inference
(inference.py:110-149 builds product>>product).  Everything is generated on CPU with an explicit
torch.Generator so that the same call gives the same graphs here, on the GPU box and in the
golden-vector script.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

ATOM_DECODER = ['none', 'O', 'C', 'N', 'I', 'Cl', 'Si', 'F', 'Br', 'N+1', 'O-1', 'S', 'B', 'N-1', 'Zn+1', 'Cu',
                'Sn', 'P+1', 'Mg+1', 'C-1', 'P', 'S+1', 'S-1', 'Se', 'Zn', 'Mg', 'U', 'Au', 'SuNo']
BOND_DECODER = ['none', 'SINGLE', 'DOUBLE', 'TRIPLE', 'mol', 'within', 'across']
KX, KE = len(ATOM_DECODER), len(BOND_DECODER)
IDX_U, IDX_AU, IDX_SUNO = ATOM_DECODER.index('U'), ATOM_DECODER.index('Au'), ATOM_DECODER.index('SuNo')
NB_DUMMY = 15


def make_reaction_batch(B: int, n: int, seed: int = 1, P: Optional[int] = None, nb_dummy: int = NB_DUMMY,
                        kx: int = KX, ke: int = KE, suno_idx: Optional[int] = None, dummy_idx: Optional[int] = None
                        ) -> Dict[str, torch.Tensor]:
    """True reactions as dense one-hot tensors.

    P=None -> 'USPTO-like': P ~ clip(N(26,10), 5, Pmax) per graph (exercises padding);
    P=int  -> every graph has P product atoms ('full' case when 2P+nb_dummy+1 == n).
    kx/ke may be shrunk for tiny test configs (then suno_idx=kx-1, dummy_idx=kx-3, absorbing kx-2).
    """
    g = torch.Generator().manual_seed(seed)
    suno_idx = kx - 1 if suno_idx is None else suno_idx
    dummy_idx = kx - 3 if dummy_idx is None else dummy_idx
    n_real_atom = dummy_idx - 1                      # classes 1 .. dummy_idx-1 are real atoms
    Pmax = (n - nb_dummy - 1) // 2
    assert Pmax >= 1, 'n too small for the layout'
    X = torch.zeros(B, n, dtype=torch.long)
    E = torch.zeros(B, n, n, dtype=torch.long)
    node_mask = torch.zeros(B, n, dtype=torch.bool)
    am = torch.zeros(B, n, dtype=torch.long)
    mol = torch.zeros(B, n, dtype=torch.long)
    for b in range(B):
        if P is None:
            p = int(torch.clamp(torch.round(torch.randn((), generator=g) * 10 + 26), 5, Pmax).item())
            p = max(1, min(p, Pmax))
        else:
            p = P
            assert p <= Pmax
        R = p + nb_dummy
        s = R                                       # SuNo position
        prod = torch.arange(s + 1, s + 1 + p)
        atoms = torch.randint(1, n_real_atom + 1, (p,), generator=g)
        # product bonds: random spanning chain + ~10% ring closures, classes {1,2,3}
        Ep = torch.zeros(p, p, dtype=torch.long)
        order = torch.randperm(p, generator=g)
        for k in range(1, p):
            a, c = int(order[k - 1]), int(order[k])
            Ep[a, c] = Ep[c, a] = int(torch.randint(1, min(4, ke), (), generator=g))
        n_ring = max(0, p // 10)
        for _ in range(n_ring):
            a, c = [int(v) for v in torch.randint(0, p, (2,), generator=g)]
            if a != c and Ep[a, c] == 0:
                Ep[a, c] = Ep[c, a] = int(torch.randint(1, min(4, ke), (), generator=g))
        amp = torch.randperm(p, generator=g) + 1    # atom-map numbers on the product
        rperm = torch.randperm(p, generator=g)      # reactant slot k holds product atom rperm[k]
        X[b, prod] = atoms
        E[b, s + 1:s + 1 + p, s + 1:s + 1 + p] = Ep
        am[b, prod] = amp
        mol[b, prod] = 2
        X[b, s] = suno_idx
        if ke > 4:                                  # supernode 'mol' edges (add_supernode_edges=True)
            E[b, s, prod] = 4
            E[b, prod, s] = 4
        X[b, :p] = atoms[rperm]
        E[b, :p, :p] = Ep[rperm][:, rperm]
        am[b, :p] = amp[rperm]
        mol[b, :p] = 1
        X[b, p:R] = dummy_idx
        node_mask[b, :s + 1 + p] = True
    Xo = F.one_hot(X, kx).float() * node_mask[:, :, None]
    Eo = F.one_hot(E, ke).float() * (node_mask[:, :, None, None] & node_mask[:, None, :, None])
    eye = torch.eye(n, dtype=torch.bool)
    Eo[:, eye] = 0
    # encode_no_element (utils/placeholder.py:9-18): padded rows/pairs and the diagonal are class 0
    Xo[..., 0] = torch.where(Xo.sum(-1) == 0, torch.ones(()), Xo[..., 0])
    Eo[..., 0] = torch.where(Eo.sum(-1) == 0, torch.ones(()), Eo[..., 0])
    return dict(X=Xo, E=Eo, y=torch.zeros(B, 1), node_mask=node_mask, atom_map_numbers=am, mol_assignments=mol)


def noised_state(true: Dict[str, torch.Tensor], t: int, T: int, abs_x: int, seed: int = 2) -> Dict[str, torch.Tensor]:
    """State at step t of the absorbing process: reactant-side nodes absorbed to `abs_x` and
    reactant-reactant edges to class 0 with probability t/T, product side untouched."""
    g = torch.Generator().manual_seed(seed)
    X, E = true['X'].clone(), true['E'].clone()
    B, n, kx = X.shape
    ke = E.shape[-1]
    suno = X.argmax(-1) == kx - 1
    ar = torch.arange(n)[None, :].expand(B, n)
    spos = torch.where(suno, ar, torch.full_like(ar, -1)).max(-1).values
    rct = ar < spos[:, None]
    flipx = (torch.rand(B, n, generator=g) < t / T) & rct
    X[flipx] = F.one_hot(torch.tensor(abs_x), kx).float()
    u = torch.rand(B, n, n, generator=g)
    u = torch.triu(u, 1); u = u + u.transpose(1, 2)
    flipe = (u < t / T) & rct[:, :, None] & rct[:, None, :] & ~torch.eye(n, dtype=torch.bool)[None]
    E[flipe] = F.one_hot(torch.tensor(0), ke).float()
    out = dict(true); out['X'] = X; out['E'] = E
    out['y'] = torch.full((B, 1), float(t))
    return out


def shared_pos_encodings(true: Dict[str, torch.Tensor], dx: int, seed: int = 3) -> torch.Tensor:
    """N(0,1) rows of width dx shared between mapped reactant/product nodes (stands in for the
    reference's Laplacian encodings, neuralnet/transformer_model_with_y.py:221-287; they are an
    input at the nn.Module boundary)."""
    g = torch.Generator().manual_seed(seed)
    am = true['atom_map_numbers']
    B, n = am.shape
    table = torch.randn(B, int(am.max()) + 1, dx, generator=g)
    table[:, 0] = 0
    return torch.gather(table, 1, am[:, :, None].expand(-1, -1, dx)).contiguous()


def exp_noise(shape, seed: int) -> torch.Tensor:
    """Exp(1) race noise for the categorical sampler (see oracle.race_draw)."""
    g = torch.Generator().manual_seed(seed)
    return torch.empty(shape).exponential_(1.0, generator=g)
