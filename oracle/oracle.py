"""CPU oracle for the DiffAlign denoiser + reverse-diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``diffalign_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

It is a from-scratch plain-PyTorch (CPU, fp32) restatement of the reference algorithm.
Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Parity is PINNED: ``tests/golden/make_golden.py`` imports the
reference in the build container, runs it on seeded synthetic inputs and commits the
inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
file against every one of those vectors.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# a1  container semantics: PlaceHolder.mask / encode_no_element
# --------------------------------------------------------------------------------------
def encode_no_element(A: Tensor) -> Tensor:
    """utils/placeholder.py:9-18 -- any all-zero (sum == 0) feature vector gets channel 0 := 1."""
    if A.shape[-1] == 0:
        return A
    A = A.clone()
    empty = A.sum(dim=-1) == 0
    first = A[..., 0]
    first[empty] = 1
    A[..., 0] = first
    return A


def mask_dense(X: Tensor, E: Tensor, node_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """utils/placeholder.py:81-102 (collapse=False)."""
    m = node_mask.to(X.dtype)
    X = X * m[:, :, None]
    E = E * m[:, :, None, None] * m[:, None, :, None]
    n = E.shape[1]
    eye = torch.eye(n, dtype=torch.bool)
    E = E.clone()
    E[:, eye] = 0
    return encode_no_element(X), encode_no_element(E)


def collapse_dense(X: Tensor, E: Tensor, node_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """utils/placeholder.py:90-94 (collapse=True): argmax then zero masked."""
    Xi = X.argmax(-1)
    Ei = E.argmax(-1)
    Xi = Xi * node_mask.long()
    pair = node_mask[:, :, None] & node_mask[:, None, :]
    Ei = Ei * pair.long()
    return Xi, Ei


# --------------------------------------------------------------------------------------
# a2  free-to-diffuse masks (closed form of utils/graph.py:462-619)
# --------------------------------------------------------------------------------------
def free_masks(orig_X: Tensor, node_mask: Tensor, atom_map_numbers: Optional[Tensor], mode: Optional[str],
               suno_idx: int, state_idx_to_mask: Sequence[int] = (0,), include_supernode: bool = True
               ) -> Tuple[Tensor, Tensor]:
    """Per-node masks (mask_x, mask_e), True = free to diffuse.

    utils/graph.py:565-594 get_mask_product: mask[b,i] = (i <= lastSuNo[b]) and (i != n-1)
    utils/graph.py:536-563 get_mask_sn (one SuNo per graph): mask[b,i] = (i != SuNo position)
    utils/graph.py:488-493 'product_and_sn' = product * sn; :500-508 'atom_mapping'
    utils/graph.py:524-529 state mask: argmax(orig.X) not in node_idx_to_mask (always holds
    'none'=0 because get_index_from_states appends it, :679-680)
    """
    B, n, _ = orig_X.shape
    cls = orig_X.argmax(-1)
    ar = torch.arange(n)[None, :].expand(B, n)
    is_suno = cls == suno_idx
    last_suno = torch.where(is_suno, ar, torch.full_like(ar, -1)).max(-1).values  # (B,)
    m_prod = (ar <= last_suno[:, None]) & (ar != n - 1)
    if include_supernode:
        # single SuNo per graph on this path: the (last) SuNo position is excluded
        m_sn = ar != last_suno[:, None]
    else:
        m_sn = node_mask.clone()
    if mode == 'product':
        mx = me = m_prod
    elif mode == 'sn':
        mx = me = m_sn
    elif mode == 'product_and_sn':
        mx = me = m_prod & m_sn
    elif mode == 'reactant':
        mx = me = ~(m_prod & m_sn)
    elif mode == 'reactant_and_sn':
        mx = me = (~(m_prod & m_sn)) & m_sn
    elif mode == 'atom_mapping':
        me = m_prod & m_sn
        mx = (atom_map_numbers == 0) & me
    else:
        mx = me = node_mask.clone()
    st = torch.ones_like(mx)
    for s in state_idx_to_mask:
        st &= cls != s
    return mx & st, me & st


def expand_masks(mask_x: Tensor, mask_e: Tensor, Kx: int, Ke: int) -> Tuple[Tensor, Tensor]:
    """utils/graph.py:607-619: mask_X[b,i,:]=mask_x[b,i]; mask_E[b,i,j,:]=mask_e[b,i]&mask_e[b,j]."""
    mX = mask_x[:, :, None].expand(-1, -1, Kx)
    mE = (mask_e[:, :, None] & mask_e[:, None, :])[..., None].expand(-1, -1, -1, Ke)
    return mX, mE


def apply_mask(orig_X: Tensor, orig_E: Tensor, z_X: Tensor, z_E: Tensor, mask_x: Tensor, mask_e: Tensor,
               as_logits: bool = False) -> Tuple[Tensor, Tensor]:
    """utils/graph.py:440-455: where not free take orig (x100 if as_logits) else z."""
    scale = 100.0 if as_logits else 1.0
    mX = mask_x[:, :, None]
    mE = (mask_e[:, :, None] & mask_e[:, None, :])[..., None]
    return torch.where(mX, z_X, orig_X * scale), torch.where(mE, z_E, orig_E * scale)


# --------------------------------------------------------------------------------------
# a5  transition tables
# --------------------------------------------------------------------------------------
def cosine_betas_discrete(timesteps: int, s: float = 0.0) -> np.ndarray:
    """utils/diffusion/helpers.py:634-643 (called with s=0, noise_schedule.py:53)."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    return (1 - ac[1:] / ac[:-1]).squeeze()


def _fix_masked_states(Q: Tensor, idx: Optional[Sequence[int]]) -> Tensor:
    """noise_schedule.py:98-121: zero rows/cols of masked states, diagonal 1, renormalise rows."""
    Q = Q.clone()
    if idx is not None:
        idx = list(idx)
        Q[..., :, idx] = 0.
        Q[..., idx, :] = 0.
        Q[..., idx, idx] = 1.
    return Q / Q.sum(-1, keepdim=True)


def _finish(tables: Dict[str, Tensor], x_limit: Tensor, e_limit: Tensor, y_limit: Tensor,
            node_idx_to_mask, edge_idx_to_mask) -> Dict[str, Tensor]:
    out = dict(
        Qt_x=_fix_masked_states(tables['Qt_x'], node_idx_to_mask),
        Qt_e=_fix_masked_states(tables['Qt_e'], edge_idx_to_mask),
        Qtb_x=_fix_masked_states(tables['Qtb_x'], node_idx_to_mask),
        Qtb_e=_fix_masked_states(tables['Qtb_e'], edge_idx_to_mask),
        Qt_y=tables['Qt_y'], Qtb_y=tables['Qtb_y'],
    )
    x_limit = x_limit.clone(); e_limit = e_limit.clone()
    if node_idx_to_mask is not None:
        x_limit[list(node_idx_to_mask)] = 0.
    if edge_idx_to_mask is not None:
        e_limit[list(edge_idx_to_mask)] = 0.
    out['x_limit'] = x_limit / x_limit.sum()
    out['e_limit'] = e_limit / e_limit.sum()
    out['y_limit'] = y_limit
    return out


def absorbing_tables(T: int, Kx: int, Ke: int, Ky: int, abs_x: int, abs_e: int = 0,
                     node_idx_to_mask=None, edge_idx_to_mask=None) -> Dict[str, Tensor]:
    """noise_schedule.py:294-326: Q_t=(1-b)I+b*1 e_abs^T, b=1/(T-t+1); Qbar_t=(1-t/T)I+(t/T)*1 e_abs^T."""
    def fam(K, a):
        eye = torch.eye(K)
        absorb = torch.zeros(K, K); absorb[:, a] = 1.
        Qt = [eye] + [1 / (T - t + 1) * absorb + (1 - 1 / (T - t + 1)) * eye for t in range(1, T + 1)]
        Qb = [eye] + [t / T * absorb + (1 - t / T) * eye for t in range(1, T + 1)]
        return torch.stack(Qt), torch.stack(Qb)
    Qt_x, Qtb_x = fam(Kx, abs_x % Kx)
    Qt_e, Qtb_e = fam(Ke, abs_e % Ke)
    Qt_y, Qtb_y = fam(Ky + 1, Ky)
    xl = torch.zeros(Kx); xl[abs_x % Kx] = 1.
    el = torch.zeros(Ke); el[abs_e % Ke] = 1.
    yl = torch.zeros(Ky + 1); yl[Ky] = 1.
    return _finish(dict(Qt_x=Qt_x, Qtb_x=Qtb_x, Qt_e=Qt_e, Qtb_e=Qtb_e, Qt_y=Qt_y, Qtb_y=Qtb_y),
                   xl, el, yl, node_idx_to_mask, edge_idx_to_mask)


def _beta_alpha_bar(T: int) -> Tuple[Tensor, Tensor]:
    """noise_schedule.py:63-70."""
    betas = torch.from_numpy(cosine_betas_discrete(T, s=0)).float()
    alphas = 1 - torch.clamp(betas, min=0, max=0.9999)
    alphas_bar = torch.exp(torch.cumsum(torch.log(alphas), dim=0))
    return betas, alphas_bar


def marginal_tables(T: int, x_marginals: Tensor, e_marginals: Tensor, Ky: int,
                    node_idx_to_mask=None, edge_idx_to_mask=None) -> Dict[str, Tensor]:
    """noise_schedule.py:234-292: Q_t=(1-b_t)I+b_t 1 m^T; Qbar_t=abar_t I+(1-abar_t) 1 m^T."""
    betas, abar = _beta_alpha_bar(T)
    def fam(u):
        K = u.shape[-1]
        eye = torch.eye(K)
        Qt = torch.stack([betas[t] * u + (1 - betas[t]) * eye for t in range(T + 1)])
        Qb = torch.stack([abar[t] * eye + (1 - abar[t]) * u for t in range(T + 1)])
        return Qt, Qb
    ux = x_marginals.float()[None, :].expand(len(x_marginals), -1)
    ue = e_marginals.float()[None, :].expand(len(e_marginals), -1)
    uy = torch.ones(Ky, Ky) / max(Ky, 1)
    Qt_x, Qtb_x = fam(ux); Qt_e, Qtb_e = fam(ue); Qt_y, Qtb_y = fam(uy)
    return _finish(dict(Qt_x=Qt_x, Qtb_x=Qtb_x, Qt_e=Qt_e, Qtb_e=Qtb_e, Qt_y=Qt_y, Qtb_y=Qtb_y),
                   x_marginals.float().clone(), e_marginals.float().clone(), torch.ones(Ky) / max(Ky, 1),
                   node_idx_to_mask, edge_idx_to_mask)


def uniform_tables(T: int, Kx: int, Ke: int, Ky: int, node_idx_to_mask=None, edge_idx_to_mask=None):
    """noise_schedule.py:171-232."""
    return marginal_tables(T, torch.ones(Kx) / Kx, torch.ones(Ke) / Ke, Ky, node_idx_to_mask, edge_idx_to_mask)


# --------------------------------------------------------------------------------------
# a14 posterior p(z_s | z_t) ; a15 sampling ; a6 initial noise ; a17 forward noising
# --------------------------------------------------------------------------------------
def _over0_posterior(z_t: Tensor, Qt: Tensor, Qsb: Tensor, Qtb: Tensor) -> Tensor:
    """utils/diffusion/helpers.py:785-807.  z_t (B,N,K) one-hot; returns (B,N,K0,Ks)."""
    a = z_t @ Qt.transpose(-1, -2)                         # (B,N,Ks): (x_t Q_t^T)[k]
    num = a.unsqueeze(2) * Qsb.unsqueeze(1)                 # (B,N,K0,Ks)
    den = (Qtb @ z_t.transpose(-1, -2)).transpose(-1, -2)   # (B,N,K0)
    den = den.clone()
    den[den == 0] = 1e-6
    return num / den.unsqueeze(-1)


def posterior_probs(logits_X: Tensor, logits_E: Tensor, zt_X: Tensor, zt_E: Tensor, node_mask: Tensor,
                    Qt_x, Qsb_x, Qtb_x, Qt_e, Qsb_e, Qtb_e) -> Tuple[Tensor, Tensor]:
    """helpers.py:193-229 get_p_zs_given_zt(return_prob=True).  Q* are per-graph (B,K,K)."""
    B, n, Kx = zt_X.shape
    Ke = zt_E.shape[-1]
    pX = F.softmax(logits_X, -1)
    pE = F.softmax(logits_E, -1).reshape(B, n * n, Ke)
    wX = _over0_posterior(zt_X.float(), Qt_x, Qsb_x, Qtb_x)
    wE = _over0_posterior(zt_E.reshape(B, n * n, Ke).float(), Qt_e, Qsb_e, Qtb_e)
    uX = (pX.unsqueeze(-1) * wX).sum(2)
    uX[uX.sum(-1) == 0] = 1e-5
    uX = uX / uX.sum(-1, keepdim=True)
    uE = (pE.unsqueeze(-1) * wE).sum(2)
    uE[uE.sum(-1) == 0] = 1e-5
    uE = (uE / uE.sum(-1, keepdim=True)).reshape(B, n, n, Ke)
    return mask_dense(uX, uE, node_mask)


def race_draw(p: Tensor, q: Tensor) -> Tensor:
    """torch.multinomial(1) on CPU == argmax(p / q) with q ~ Exp(1) drawn from the same generator
    state (verified in tests/golden/make_golden.py); first index wins ties."""
    return (p / q).argmax(-1)


def sample_discrete(probX: Tensor, probE: Tensor, node_mask: Tensor, qX: Tensor, qE: Tensor
                    ) -> Tuple[Tensor, Tensor]:
    """helpers.py:706-752 with explicit Exp(1) noise.  Returns class indices X (B,n), E (B,n,n)."""
    B, n, Kx = probX.shape
    Ke = probE.shape[-1]
    pX = probX.clone()
    pX[~node_mask] = 1 / Kx
    Xs = race_draw(pX, qX) * node_mask.long()
    pair = node_mask[:, :, None] & node_mask[:, None, :]
    pE = probE.clone()
    pE[~pair] = 1 / Ke
    pE[:, torch.eye(n, dtype=torch.bool)] = 1 / Ke
    Es = race_draw(pE, qE)
    Es = torch.triu(Es, diagonal=1)
    Es = Es + Es.transpose(1, 2)
    Es = Es * pair.long()
    return Xs, Es


def one_hot_pair(Xs: Tensor, Es: Tensor, Kx: int, Ke: int) -> Tuple[Tensor, Tensor]:
    return F.one_hot(Xs, Kx).float(), F.one_hot(Es, Ke).float()


def sample_from_limit(x_limit: Tensor, e_limit: Tensor, node_mask: Tensor, qX: Tensor, qE: Tensor):
    """helpers.py:831-859: z_T ~ limit dist per node/pair, upper triangle mirrored, .mask()."""
    B, n = node_mask.shape
    Kx, Ke = x_limit.numel(), e_limit.numel()
    Xs = race_draw(x_limit[None, None, :].expand(B, n, -1), qX)
    Es = race_draw(e_limit[None, None, None, :].expand(B, n, n, -1), qE)
    X = F.one_hot(Xs, Kx).float()
    E = F.one_hot(Es, Ke).float()
    up = torch.triu(torch.ones(n, n), diagonal=1)[None, :, :, None]
    E = E * up
    E = E + E.transpose(1, 2)
    return mask_dense(X, E, node_mask)


def noise_probs(X: Tensor, E: Tensor, node_mask: Tensor, Qtb_x: Tensor, Qtb_e: Tensor):
    """diffusion_abstract.py:253-257: probX = X @ Qbar_t.X, probE = E @ Qbar_t.E, .mask()."""
    pX = X @ Qtb_x
    pE = E @ Qtb_e.unsqueeze(1)
    return mask_dense(pX, pE, node_mask)


# --------------------------------------------------------------------------------------
# a9-a12 denoiser forward (functional, reference state-dict key names)
# --------------------------------------------------------------------------------------
def _lin(sd, key, x):
    return F.linear(x, sd[key + '.weight'], sd[key + '.bias'])


def _stats4(T: Tensor, dims) -> Tensor:
    """neuralnet/layers.py:15-47 Xtoy/Etoy pooled statistics [mean, min, max, std(unbiased)]."""
    m = T.mean(dim=dims)
    mi = T.amin(dim=dims)
    ma = T.amax(dim=dims)
    sd = T.std(dim=dims)
    return torch.cat([m, mi, ma, sd], dim=-1)


def node_edge_block(sd: Dict[str, Tensor], p: str, X, E, y, node_mask, n_head: int):
    """neuralnet/layers.py:204-284 NodeEdgeBlock.forward (p = key prefix '...self_attn')."""
    B, n, dx = X.shape
    df = dx // n_head
    xm = node_mask.to(X.dtype)[:, :, None]
    em = xm[:, :, None, :] * xm[:, None, :, :]
    Q = _lin(sd, p + '.q', X) * xm
    K = _lin(sd, p + '.k', X) * xm
    V = _lin(sd, p + '.v', X) * xm
    Y = Q[:, :, None, :] * K[:, None, :, :] / math.sqrt(df)          # (B,i,j,dx)
    E1 = _lin(sd, p + '.e_mul', E) * em
    E2 = _lin(sd, p + '.e_add', E) * em
    Y = Y * (E1 + 1) + E2
    ye1 = _lin(sd, p + '.y_e_add', y)[:, None, None, :]
    ye2 = _lin(sd, p + '.y_e_mul', y)[:, None, None, :]
    newE = _lin(sd, p + '.e_out', ye1 + (ye2 + 1) * Y) * em
    attn = F.softmax(Y, dim=2)                                       # over keys j, per channel, no mask
    wV = (attn * V[:, None, :, :]).sum(dim=2)
    yx1 = _lin(sd, p + '.y_x_add', y)[:, None, :]
    yx2 = _lin(sd, p + '.y_x_mul', y)[:, None, :]
    newX = _lin(sd, p + '.x_out', yx1 + (yx2 + 1) * wV) * xm
    x_y = F.linear(_stats4(X, 1), sd[p + '.x_y.lin.weight'], sd[p + '.x_y.lin.bias'])
    e_y = F.linear(_stats4(E, (1, 2)), sd[p + '.e_y.lin.weight'], sd[p + '.e_y.lin.bias'])
    new_y = y + x_y + e_y
    new_y = _lin(sd, p + '.y_out.2', F.relu(_lin(sd, p + '.y_out.0', new_y)))
    return newX, newE, new_y


def node_edge_block_improved(sd: Dict[str, Tensor], p: str, X, E, y, node_mask, n_head: int):
    """neuralnet/layers.py:87-164 NodeEdgeBlockImproved.forward: newE = e_out(Y) without the y FiLM; one softmax per head over the
    head's summed scores."""
    B, n, dx = X.shape
    df = dx // n_head
    xm = node_mask.to(X.dtype)[:, :, None]
    em = xm[:, :, None, :] * xm[:, None, :, :]
    Q = _lin(sd, p + '.q', X) * xm
    K = _lin(sd, p + '.k', X) * xm
    V = _lin(sd, p + '.v', X) * xm
    Y = Q[:, :, None, :] * K[:, None, :, :] / math.sqrt(df)
    Y = Y * (_lin(sd, p + '.e_mul', E) * em + 1) + _lin(sd, p + '.e_add', E) * em
    newE = _lin(sd, p + '.e_out', Y) * em
    attn = F.softmax(Y.reshape(B, n, n, n_head, df).sum(-1), dim=2)
    wV = (attn[..., None] * V.reshape(B, 1, n, n_head, df)).sum(dim=2).reshape(B, n, dx)
    yx1 = _lin(sd, p + '.y_x_add', y)[:, None, :]
    yx2 = _lin(sd, p + '.y_x_mul', y)[:, None, :]
    newX = _lin(sd, p + '.x_out', yx1 + (yx2 + 1) * wV) * xm
    x_y = F.linear(_stats4(X, 1), sd[p + '.x_y.lin.weight'], sd[p + '.x_y.lin.bias'])
    e_y = F.linear(_stats4(E, (1, 2)), sd[p + '.e_y.lin.weight'], sd[p + '.e_y.lin.bias'])
    new_y = _lin(sd, p + '.y_out.2', F.relu(_lin(sd, p + '.y_out.0', y + x_y + e_y)))
    return newX, newE, new_y


def _ln(sd, key, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[key + '.weight'], sd[key + '.bias'], eps)


def xey_layer(sd, p: str, X, E, y, node_mask, n_head: int, improved: bool = False):
    """neuralnet/layers.py:330-361 XEyTransformerLayer.forward in eval mode (dropout = identity)."""
    nX, nE, ny = (node_edge_block_improved if improved else node_edge_block)(sd, p + '.self_attn', X, E, y, node_mask, n_head)
    X = _ln(sd, p + '.normX1', X + nX)
    E = _ln(sd, p + '.normE1', E + nE)
    y = _ln(sd, p + '.norm_y1', y + ny)
    X = _ln(sd, p + '.normX2', X + _lin(sd, p + '.linX2', F.relu(_lin(sd, p + '.linX1', X))))
    E = _ln(sd, p + '.normE2', E + _lin(sd, p + '.linE2', F.relu(_lin(sd, p + '.linE1', E))))
    y = _ln(sd, p + '.norm_y2', y + _lin(sd, p + '.lin_y2', F.relu(_lin(sd, p + '.lin_y1', y))))
    return X, E, y


def align_index(atom_map_numbers: Tensor, mol_assignments: Tensor) -> Tensor:
    """Product->reactant alignment as a gather index (semantics of
    transformer_model_with_y.py:506-555 with alignment_type='correct' and
    utils/math_utils.py:3-44): src[b,r] = product node carrying the same atom-map number as
    reactant-side node r, or -1 when r is not an atom-mapped reactant-side node."""
    B, n = atom_map_numbers.shape
    prod_id = mol_assignments.max(-1).values
    is_prod = mol_assignments == prod_id[:, None]
    am_prod = torch.where(is_prod, atom_map_numbers, torch.zeros_like(atom_map_numbers))
    am_rct = torch.where(is_prod, torch.zeros_like(atom_map_numbers), atom_map_numbers)
    src = torch.full((B, n), -1, dtype=torch.long)
    for b in range(B):
        pos = {int(a): i for i, a in enumerate(am_prod[b].tolist()) if a > 0}
        for r, a in enumerate(am_rct[b].tolist()):
            if a > 0:
                assert a in pos, 'reactant atom-map number without a product partner'
                src[b, r] = pos[a]
        assert int((am_rct[b] > 0).sum()) == len(pos)   # transformer_model_with_y.py:523
    return src


def align_index_old(atom_map_numbers: Tensor, mol_assignments: Tensor) -> Tensor:
    """alignment_type='old' (transformer_model_with_y.py:526-538): create_permutation_matrix_torch (utils/math_utils.py:3-44) matches
    the two lists rank by rank, and both lists (node indices minus their minimum) are increasing, so the k-th atom-mapped
    reactant-side node takes the k-th atom-mapped product node."""
    B, n = atom_map_numbers.shape
    is_prod = mol_assignments == mol_assignments.max(-1).values[:, None]
    src = torch.full((B, n), -1, dtype=torch.long)
    for b in range(B):
        prod = [i for i in range(n) if is_prod[b, i] and atom_map_numbers[b, i] > 0]
        rct = [i for i in range(n) if not is_prod[b, i] and atom_map_numbers[b, i] > 0]
        assert len(prod) == len(rct)
        for r, p in zip(rct, prod):
            src[b, r] = p
    return src


def p_to_r_skip(X_o, E_o, X_in, E_in, src: Tensor, s1, s2, Kx: int, Ke: int):
    """transformer_model_with_y.py:686-701.  s1, s2: floats or the (1,) parameters themselves (then differentiable in them)."""
    X_o = X_o.clone(); E_o = E_o.clone()
    B, n = src.shape
    for b in range(B):
        sel = src[b] >= 0
        idx = src[b][sel]
        X_o[b, sel] += X_in[b, idx, :Kx] * s1
        pair = sel[:, None] & sel[None, :]
        Ea = E_in[b, :, :, :Ke][idx][:, idx]                          # (m,m,Ke)
        E_o[b][pair] += Ea.reshape(-1, Ke) * s1
        onehot0 = torch.zeros(Ke); onehot0[0] = 1.
        E_o[b][~pair] += onehot0 * s2
    return X_o, E_o


def aligned_inputs(X_in, E_in, src: Tensor, Kx: int, Ke: int):
    """transformer_model_with_y.py:630-645 (input_alignment): X_to_concatenate / E_to_concatenate."""
    B, n = src.shape
    Xc, Ec = torch.zeros(B, n, Kx), torch.zeros(B, n, n, Ke)
    onehot0 = torch.zeros(Ke); onehot0[0] = 1.
    for b in range(B):
        sel = src[b] >= 0
        idx = src[b][sel]
        Xc[b, sel] = X_in[b, idx, :Kx]
        pair = sel[:, None] & sel[None, :]
        Ec[b][pair] = E_in[b, :, :, :Ke][idx][:, idx].reshape(-1, Ke)
        Ec[b][~pair] = onehot0
    return Xc, Ec


def denoiser_forward(sd: Dict[str, Tensor], X, E, y, node_mask, n_layers: int, n_head: int,
                     out_dims: Tuple[int, int, int], atom_map_numbers=None, pos_encodings=None,
                     mol_assignments=None, p_to_r_skip_connection=False, prefix: str = '', input_alignment=False, alignment_type='correct', improved=False):
    """transformer_model_with_y.py:620-713 (atom-map variant when pos_encodings is given) and
    :74-111 (plain GraphTransformerWithY).  Returns un-masked logits (X, E, y)."""
    Kx, Ke, Ky = out_dims
    B, n = X.shape[:2]
    align = align_index if alignment_type == 'correct' else align_index_old
    if input_alignment:
        Xc, Ec = aligned_inputs(X, E, align(atom_map_numbers, mol_assignments), Kx, Ke)
        X, E = torch.cat([X, Xc], -1), torch.cat([E, Ec], -1)
    X_in, E_in = X, E
    def mlp_in(name, t):
        return F.relu(_lin(sd, f'{prefix}{name}.2', F.relu(_lin(sd, f'{prefix}{name}.0', t))))
    Eh = mlp_in('mlp_in_E', E)
    Eh = (Eh + Eh.transpose(1, 2)) / 2
    Xh, Eh = mask_dense(mlp_in('mlp_in_X', X), Eh, node_mask)       # encode_no_element on hidden feats
    yh = mlp_in('mlp_in_y', y)
    if pos_encodings is not None:
        Xh = Xh + pos_encodings
    for l in range(n_layers):
        Xh, Eh, yh = xey_layer(sd, f'{prefix}tf_layers.{l}', Xh, Eh, yh, node_mask, n_head, improved)
    Xo = _lin(sd, prefix + 'mlp_out_X.2', F.relu(_lin(sd, prefix + 'mlp_out_X.0', Xh))) + X_in[..., :Kx]
    Eo = _lin(sd, prefix + 'mlp_out_E.2', F.relu(_lin(sd, prefix + 'mlp_out_E.0', Eh))) + E_in[..., :Ke]
    Eo = Eo * (~torch.eye(n, dtype=torch.bool))[None, :, :, None]
    yo = yh[..., :Ky] + y[..., :Ky]
    Eo = 0.5 * (Eo + Eo.transpose(1, 2))
    if p_to_r_skip_connection and input_alignment:              # :702-704: the concatenated tensors again, ALL of the edge part times s2
        Xo = Xo + Xc * sd[prefix + 'skip_scaling']
        Eo = Eo + Ec * sd[prefix + 'skip_scaling_2']
    elif p_to_r_skip_connection:
        src = align(atom_map_numbers, mol_assignments)
        Xo, Eo = p_to_r_skip(Xo, Eo, X_in, E_in, src, sd[prefix + 'skip_scaling'], sd[prefix + 'skip_scaling_2'], Kx, Ke)
    return Xo, Eo, yo


# --------------------------------------------------------------------------------------
# a16 one reverse step, a18-a19 ELBO terms
# --------------------------------------------------------------------------------------
def reverse_step(logits_X, logits_E, zt_X, zt_E, node_mask, tabs: Dict[str, Tensor], t: int,
                 orig_X, orig_E, mask_x, mask_e, qX, qE, temp_x=1.0, temp_e=1.0):
    """diffusion/sampling.py:82-86,165-178: log_softmax*temperature -> posterior -> draw -> apply_mask.
    Returns (prob_X, prob_E, z_s X one-hot, z_s E one-hot)."""
    B = zt_X.shape[0]
    Kx, Ke = zt_X.shape[-1], zt_E.shape[-1]
    lX = torch.log_softmax(logits_X, -1) * temp_x
    lE = torch.log_softmax(logits_E, -1) * temp_e
    ex = lambda T_, i: T_[i][None].expand(B, -1, -1)
    pX, pE = posterior_probs(lX, lE, zt_X, zt_E, node_mask,
                             ex(tabs['Qt_x'], t), ex(tabs['Qtb_x'], t - 1), ex(tabs['Qtb_x'], t),
                             ex(tabs['Qt_e'], t), ex(tabs['Qtb_e'], t - 1), ex(tabs['Qtb_e'], t))
    Xs, Es = sample_discrete(pX, pE, node_mask, qX, qE)
    Xs1, Es1 = one_hot_pair(Xs, Es, Kx, Ke)
    zX, zE = apply_mask(orig_X, orig_E, Xs1, Es1, mask_x, mask_e)
    return pX, pE, zX, zE


def posterior_true_or_pred(M: Tensor, M_t: Tensor, Qt: Tensor, Qsb: Tensor, log_space: bool = False) -> Tensor:
    """helpers.py:754-783 compute_posterior_distribution. M, M_t: (B,N,K)."""
    if not log_space:
        num = (M_t @ Qt.transpose(-1, -2)) * (M @ Qsb)
        return num / num.sum(-1, keepdim=True)
    eps = 1e-30
    Ml = torch.log_softmax(M, -1)
    ln = torch.log(M_t @ Qt.transpose(-1, -2) + eps) + torch.log(torch.exp(Ml) @ Qsb + eps)
    return ln - torch.logsumexp(ln, -1, keepdim=True)


def kl_rows(target: Tensor, pred_log: Tensor) -> Tensor:
    """F.kl_div(input=pred_log, target=target, reduction='none').sum(-1) (elbo.py:153-157)."""
    return F.kl_div(pred_log, target, reduction='none').sum(-1)


def compute_Lt(true_X, true_E, zt_X, zt_E, logits_X, logits_E, tabs, t_vec: Tensor, mask_x, mask_e, eps=1e-6,
               log_space=False):
    """diffusion/elbo.py:85-162.  t_vec (B,) integer step per graph.  Returns KL maps X (B,n), E (B,n,n)."""
    B, n, Kx = true_X.shape
    Ke = true_E.shape[-1]
    ti = t_vec.long()
    g = lambda name, idx: tabs[name][idx]
    Qt_x, Qsb_x = g('Qt_x', ti), g('Qtb_x', ti - 1)
    Qt_e, Qsb_e = g('Qt_e', ti), g('Qtb_e', ti - 1)
    fl = lambda T_: T_.reshape(B, n * n, Ke)
    qX = posterior_true_or_pred(true_X, zt_X, Qt_x, Qsb_x)
    qE = posterior_true_or_pred(fl(true_E), fl(zt_E), Qt_e, Qsb_e).reshape(B, n, n, Ke)
    qX, qE = apply_mask(true_X, true_E, qX, qE, mask_x, mask_e)
    if not log_space:
        pX = posterior_true_or_pred(F.softmax(logits_X, -1), zt_X, Qt_x, Qsb_x)
        pE = posterior_true_or_pred(fl(F.softmax(logits_E, -1)), fl(zt_E), Qt_e, Qsb_e).reshape(B, n, n, Ke)
        pX, pE = apply_mask(true_X, true_E, pX, pE, mask_x, mask_e)
        return kl_rows(qX, (pX + eps).log()), kl_rows(qE, (pE + eps).log())
    lpX = posterior_true_or_pred(logits_X, zt_X, Qt_x, Qsb_x, log_space=True)
    lpE = posterior_true_or_pred(fl(logits_E), fl(zt_E), Qt_e, Qsb_e, log_space=True).reshape(B, n, n, Ke)
    lpX, lpE = apply_mask(true_X, true_E, lpX, lpE, mask_x, mask_e, as_logits=True)
    return kl_rows(qX, torch.log_softmax(lpX, -1)), kl_rows(qE, torch.log_softmax(lpE, -1))


def kl_prior(true_X, true_E, tabs, T: int, mask_x, mask_e, eps=1e-6):
    """diffusion/elbo.py:197-238 + helpers.py:115-137."""
    B, n, Kx = true_X.shape
    Ke = true_E.shape[-1]
    pX = true_X @ tabs['Qtb_x'][T]
    pE = true_E @ tabs['Qtb_e'][T]
    lX = tabs['x_limit'][None, None, :].expand(B, n, -1)
    lE = tabs['e_limit'][None, None, None, :].expand(B, n, n, -1)
    lX, lE = apply_mask(true_X, true_E, lX, lE, mask_x, mask_e)
    pX, pE = apply_mask(true_X, true_E, pX, pE, mask_x, mask_e)
    return kl_rows(lX, (pX + eps).log()), kl_rows(lE, (pE + eps).log())


def reconstruction_logp(true_X, true_E, logits_X, logits_E, mask_x, mask_e):
    """diffusion/elbo.py:182-193 + helpers.py:139-151 (logits already apply_mask(as_logits=True)-ed here)."""
    lX, lE = apply_mask(true_X, true_E, logits_X, logits_E, mask_x, mask_e, as_logits=True)
    return (true_X * F.log_softmax(lX, -1)).sum(-1), (true_E * F.log_softmax(lE, -1)).sum(-1)


def mean_without_masked(tX: Tensor, tE: Tensor, mask_x: Tensor, mask_e: Tensor, avg_over_batch=False,
                        diffuse_nodes=True, diffuse_edges=True):
    """helpers.py:59-106.  tX (B,n), tE (B,n,n); masks per node; E mask = outer product."""
    mx = mask_x.to(tX.dtype)
    me = (mask_e[:, :, None] & mask_e[:, None, :]).to(tE.dtype).flatten(1)
    sX = tX * mx
    sE = tE.flatten(1) * me
    if avg_over_batch:
        mX, mE = sX.sum() / mx.sum(), sE.sum() / me.sum()
    else:
        mX, mE = sX.sum(-1) / mx.sum(-1), sE.sum(-1) / me.sum(-1)
    if diffuse_nodes and diffuse_edges:
        return mX + mE
    return mE if diffuse_edges else mX


def ce_loss(logits_X, logits_E, true_X, true_E, mask_x, mask_e, lambda_E: float):
    """helpers.py:153-191 with masks from apply_mask (diffusion_abstract.py:198-207)."""
    me = (mask_e[:, :, None] & mask_e[:, None, :])
    lX = F.cross_entropy(logits_X[mask_x], true_X[mask_x].argmax(-1), reduction='none')
    lE = F.cross_entropy(logits_E[me], true_E[me].argmax(-1), reduction='none')
    return lX, lE, lX.mean() + lambda_E * lE.mean()


# --------------------------------------------------------------------------------------
# a16 full reverse-diffusion chain (diffusion/sampling.py:11-197) with explicit noise
# --------------------------------------------------------------------------------------
def elbo_batch_quick(true_X, true_E, zt_X, zt_E, logits_X, logits_E, tabs, t_vec: Tensor, T: int, mask_x, mask_e,
                     lambda_E: float, diffuse_nodes=True, diffuse_edges=True):
    """diffusion/elbo.py:273-334 (avg_over_batch=True): the single-timestep ELBO estimator the 'vb' / 'vbce' training losses
    use -- log-space L_t rows, replaced by the reconstruction term on graphs with t == 1, masked mean, times T, plus the prior KL."""
    LtX, LtE = compute_Lt(true_X, true_E, zt_X, zt_E, logits_X, logits_E, tabs, t_vec, mask_x, mask_e, log_space=True)
    kpX, kpE = kl_prior(true_X, true_E, tabs, T, mask_x, mask_e)
    r1X, r1E = reconstruction_logp(true_X, true_E, F.log_softmax(logits_X, -1), F.log_softmax(logits_E, -1), mask_x, mask_e)
    is1 = (t_vec.long() == 1)
    tX = torch.where(is1[:, None], -r1X, LtX)
    tE = torch.where(is1[:, None, None], -r1E, LtE) * lambda_E
    kw = dict(avg_over_batch=True, diffuse_nodes=diffuse_nodes, diffuse_edges=diffuse_edges)
    return mean_without_masked(tX, tE, mask_x, mask_e, **kw) * T + mean_without_masked(kpX, kpE * lambda_E, mask_x, mask_e, **kw)


def fix_masks(B: int, n: int, node_idx, edge_idx) -> Tuple[Tensor, Tensor]:
    """utils/graph.py:621-656 fix_nodes_and_edges_by_idx: (B,n) / (B,n,n) bool masks from per-graph index lists."""
    mX = torch.zeros(B, n, dtype=torch.bool); mE = torch.zeros(B, n, n, dtype=torch.bool)
    if node_idx is not None:
        for b in range(B):
            mX[b, node_idx[b]] = True
    if edge_idx is not None:
        for b in range(B):
            for i, j in edge_idx[b]:
                mE[b, i, j] = True; mE[b, j, i] = True
    return mX, mE


def duplicate(d: Dict[str, Tensor], n_samples: int, mask_: bool = True) -> Dict[str, Tensor]:
    """utils/graph.py:130-158 duplicate_data: .mask() then repeat_interleave of every field along the batch axis."""
    d = dict(d)
    if mask_:
        d['X'], d['E'] = mask_dense(d['X'], d['E'], d['node_mask'])
    return {k: (v.repeat_interleave(n_samples, dim=0) if torch.is_tensor(v) else v) for k, v in d.items()}


def to_dense(x: Tensor, batch: Tensor, edge_index: Tensor, edge_attr: Tensor, atom_map: Optional[Tensor] = None,
             mol: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """utils/graph.py:189-210 to_dense: to_dense_batch (rows of graph b, in order, padded to the largest graph),
    remove_self_loops, to_dense_adj (adj[b, i - ptr_b, j - ptr_b] += edge_attr), encode_no_element on X and E."""
    B = int(batch.max()) + 1
    counts = torch.bincount(batch, minlength=B)
    n = int(counts.max())
    ptr = torch.cumsum(counts, 0) - counts
    pos = torch.arange(batch.numel()) - ptr[batch]
    X = x.new_zeros(B, n, x.shape[1]); X[batch, pos] = x
    nm = torch.zeros(B, n, dtype=torch.bool); nm[batch, pos] = True
    keep = edge_index[0] != edge_index[1]
    ei, ea = edge_index[:, keep], edge_attr[keep]
    E = edge_attr.new_zeros(B, n, n, edge_attr.shape[1])
    b = batch[ei[0]]
    E.index_put_((b, ei[0] - ptr[b], ei[1] - ptr[b]), ea, accumulate=True)
    out = dict(X=encode_no_element(X), E=encode_no_element(E), node_mask=nm)
    for name, v in (('atom_map_numbers', atom_map), ('mol_assignments', mol)):
        if v is not None:
            d_ = v.new_zeros(B, n); d_[batch, pos] = v
            out[name] = d_
    return out


def counter_exp_noise(seeds: Tensor, call: int, n: int, kx: int, ke: int) -> Tuple[Tensor, Tensor]:
    """Exp(1) noise as a function of (chain seed, draw call, element): the integers of csrc/k_graph.hip counter_exp (numpy
    uint64, wrap-around arithmetic); no counterpart in the reference."""
    def stream(st, per):
        with np.errstate(over='ignore'):
            sd = seeds.numpy().astype(np.int64).view(np.uint64)[:, None]
            z = sd + np.uint64(0x9E3779B97F4A7C15) * np.uint64(st + 1)
            z = z ^ (np.arange(per, dtype=np.uint64)[None, :] * np.uint64(0xD1B54A32D192ED03))
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        u = ((z >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        return torch.from_numpy(-np.log(u))
    B = seeds.shape[0]
    return stream(2 * call, n * kx).reshape(B, n, kx), stream(2 * call + 1, n * n * ke).reshape(B, n, n, ke)


def guided_probs(pX, pE, uX, uE, w: float, eps: float = 1e-30):
    """diffusion/sampling.py:108-113: classifier-free guidance on the two posteriors."""
    gX = torch.softmax((1 + w) * torch.log(pX + eps) - w * torch.log(uX + eps), -1)
    gE = torch.softmax((1 + w) * torch.log(pE + eps) - w * torch.log(uE + eps), -1)
    gX[gX < 1e-28] = 0.
    gE[gE < 1e-28] = 0.
    return gX, gE


def sample_chain(forward_fn, true: Dict[str, Tensor], tabs_eval: Dict[str, Tensor], T: int, T_eval: int, mask_mode: str,
                 suno_idx: int, noise_fn, temp=(1.0, 1.0), node_idx=None, edge_idx=None, guidance=None):
    """forward_fn(zX, zE, y) -> (logits_X, logits_E).  noise_fn(call_index, shape) -> Exp(1) noise; calls are numbered
    in the reference's multinomial order: limit draw (X, E), then per step (X, E).  node_idx / edge_idx: inpainting lists
    (sampling.py:42-45,175-178: the listed nodes / edges are copied back from the condition after every draw).
    guidance = dict(w=..., forward_uncond=fn(zX, zE, y) -> logits, tabs_train=...): classifier-free guidance (sampling.py:88-113);
    forward_uncond sees z_t with the condition (everything fixed except the supernode) zeroed."""
    X0, E0, nm = true['X'], true['E'], true['node_mask']
    B, n, Kx = X0.shape
    Ke = E0.shape[-1]
    c = 0
    zX, zE = sample_from_limit(tabs_eval['x_limit'], tabs_eval['e_limit'], nm, noise_fn(c, (B * n, Kx)).reshape(B, n, Kx),
                               noise_fn(c + 1, (B * n * n, Ke)).reshape(B, n, n, Ke))
    c += 2
    mx, me = free_masks(X0, nm, true.get('atom_map_numbers'), mask_mode, suno_idx)
    zX, zE = apply_mask(X0, E0, zX, zE, mx, me)
    fX, fE = fix_masks(B, n, node_idx, edge_idx)

    def fix(zX, zE):
        return torch.where(fX[..., None], X0, zX), torch.where(fE[..., None], E0, zE)

    zX, zE = fix(zX, zE)
    step = T // T_eval
    for t in range(T_eval, 0, -1):
        y = torch.full((B, 1), float(t * step))
        lX, lE = forward_fn(zX, zE, y)
        qX = noise_fn(c, (B * n, Kx)).reshape(B, n, Kx)
        qE = noise_fn(c + 1, (B * n * n, Ke)).reshape(B, n, n, Ke)
        c += 2
        if guidance is None:
            _, _, zX, zE = reverse_step(lX, lE, zX, zE, nm, tabs_eval, t, X0, E0, mx, me, qX, qE, temp[0], temp[1])
        else:
            pX, pE, _, _ = reverse_step(lX, lE, zX, zE, nm, tabs_eval, t, X0, E0, mx, me, qX, qE, temp[0], temp[1])
            fx, fe = free_masks(X0, nm, true.get('atom_map_numbers'), mask_mode, suno_idx, include_supernode=False)
            MX, ME = expand_masks(fx, fe, Kx, Ke)
            uX, uE = guidance['forward_uncond'](zX * MX, zE * ME, y)
            uX, uE, _, _ = reverse_step(uX, uE, zX, zE, nm, guidance['tabs_train'], t, X0, E0, mx, me, qX, qE)
            gX, gE = guided_probs(pX, pE, uX, uE, guidance['w'])
            Xs, Es = sample_discrete(gX, gE, nm, qX, qE)
            zX, zE = apply_mask(X0, E0, *one_hot_pair(Xs, Es, Kx, Ke), mx, me)
        zX, zE = fix(zX, zE)
    return zX, zE


# --------------------------------------------------------------------------------------
# a8  extra input features (cycle counts, Laplacian eigen-features, charge / valency / weight)
# --------------------------------------------------------------------------------------
def cycle_features(E: Tensor, node_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """neuralnet/extra_features.py:75-90 (NodeCycleFeatures) + :227-298 (KNodeCycles): k-cycle counts per node
    (k = 3, 4, 5) and per graph (k = 3..6) from the powers of A = sum of the non-'none' edge channels."""
    A = E[..., 1:].sum(dim=-1).float()
    d = A.sum(dim=-1)
    k2 = A @ A; k3 = k2 @ A; k4 = k3 @ A; k5 = k4 @ A; k6 = k5 @ A                         # :234-242
    dg = lambda M: torch.diagonal(M, dim1=-2, dim2=-1)
    c3 = dg(k3)                                                                            # :244-247
    c4 = dg(k4) - d * (d - 1) - (A @ d.unsqueeze(-1)).sum(dim=-1)                          # :249-252
    tri = dg(k3)
    c5 = dg(k5) - 2 * tri * d - (A @ tri.unsqueeze(-1)).sum(dim=-1) + tri                  # :254-259
    c6 = (dg(k6).sum(-1) - 3 * dg(k3 ** 2).sum(-1) + 9 * (A * k2.pow(2)).sum(dim=[-2, -1])  # :261-277
          - 6 * (dg(k2) * dg(k4)).sum(-1) + 6 * dg(k4).sum(-1) - 4 * dg(k3).sum(-1) + 4 * dg(k2).pow(3).sum(-1)
          + 3 * k3.sum(dim=[-2, -1]) - 12 * dg(k2).pow(2).sum(-1) + 4 * dg(k2).sum(-1))
    x = torch.stack([c3 / 2, c4 / 2, c5 / 2], dim=-1).float()
    y = torch.stack([c3.sum(-1) / 6, c4.sum(-1) / 8, c5.sum(-1) / 10, c6 / 12], dim=-1).float()
    x = x * node_mask.unsqueeze(-1) / 10                                                   # :84-89
    y = y / 10
    return x.clamp(max=1), y.clamp(max=1)


def eigen_features(E: Tensor, node_mask: Tensor, pad_noise: Tensor):
    """neuralnet/extra_features.py:102-128 (EigenFeatures, mode 'all'), :133-146 (combinatorial Laplacian), :158-174
    (eigenvalue features) and :177-204 (eigenvector features).  `pad_noise` (B,n) replaces the reference's torch.randn on
    the padded slots (:189).  fp64, torch.linalg.eigh, as the reference."""
    mask = node_mask
    A = E.double()[..., 1:].sum(dim=-1) * mask.unsqueeze(1) * mask.unsqueeze(2)
    comb = torch.diag_embed(A.sum(dim=-1)) - A
    Lap = (comb + comb.transpose(1, 2)) / 2
    n = Lap.shape[-1]
    mask_diag = 2 * n * torch.eye(n, dtype=Lap.dtype).unsqueeze(0) * (~mask.unsqueeze(1)) * (~mask.unsqueeze(2))
    Lap = Lap * mask.unsqueeze(1) * mask.unsqueeze(2) + mask_diag
    evals, evecs = torch.linalg.eigh(Lap)
    evals = evals / mask.sum(dim=1, keepdim=True)
    evecs = evecs * mask.unsqueeze(2) * mask.unsqueeze(1)
    k = 5
    ncomp = (evals < 1e-5).sum(dim=-1)                                                      # :166
    ext = int(ncomp.max()) + k - n
    ev = torch.hstack((evals, 2 * torch.ones(evals.shape[0], ext, dtype=evals.dtype))) if ext > 0 else evals
    ev5 = torch.gather(ev, 1, torch.arange(k).unsqueeze(0) + ncomp.unsqueeze(1))           # :172-173
    first = torch.round(evecs[:, :, 0], decimals=3) * mask + pad_noise.to(evecs.dtype) * (~mask)   # :187-190
    mode = torch.mode(first, dim=1).values
    nonlcc = ((first != mode.unsqueeze(1)) * mask).float()                                  # :192-193
    ext = int(ncomp.max()) + 2 - n
    V = torch.cat((evecs, torch.zeros(evecs.shape[0], n, ext, dtype=evecs.dtype)), dim=2) if ext > 0 else evecs
    idx = (torch.arange(2).view(1, 1, 2) + ncomp.view(-1, 1, 1)).expand(-1, n, -1)
    vec2 = torch.gather(V, 2, idx) * mask.unsqueeze(2)                                      # :199-202
    return ncomp.float(), ev5.float(), nonlcc, vec2.float(), Lap


def molecular_features(X: Tensor, E: Tensor, valencies, bond_orders, atom_weights, max_weight: float):
    """neuralnet/extra_features_molecular.py:29-72: 'charge' (argmax index of X*valencies minus the sum of the argmax
    indices of E*bond_orders), 'valency' (sum of the argmax indices of E) and molecular weight / max_weight."""
    bo = torch.tensor(bond_orders, dtype=E.dtype).reshape(1, 1, 1, -1)
    current = (E * bo).argmax(dim=-1).sum(dim=-1)                                           # :40-42
    normal = (X * torch.tensor(valencies, dtype=X.dtype).reshape(1, 1, -1)).argmax(dim=-1)  # :44-46
    charge = (normal - current).type_as(X)
    valency = E.argmax(dim=-1).sum(dim=-1).type_as(X)                                       # :59-60
    w = torch.tensor(list(atom_weights), dtype=torch.float32)[X.argmax(dim=-1)].sum(dim=-1, keepdim=True).type_as(X) / max_weight
    return torch.stack((charge, valency), dim=-1), w


def extra_features(X: Tensor, E: Tensor, node_mask: Tensor, max_n_nodes: int, valencies, bond_orders, atom_weights,
                   max_weight: float, pad_noise: Tensor) -> Tuple[Tensor, Tensor]:
    """diffusion/diffusion_abstract.py:350-372 with ExtraFeatures('all') (extra_features.py:31-33, :56-67): the 8 extra
    node features [3 cycle counts, not-in-largest-component, 2 eigenvectors, charge, valency] and the 12 extra global
    features [n / max_n, 4 cycle counts, #components, 5 eigenvalues, weight]."""
    xc, yc = cycle_features(E, node_mask)
    ncomp, ev5, nonlcc, vec2, _ = eigen_features(E, node_mask, pad_noise)
    cv, w = molecular_features(X, E, valencies, bond_orders, atom_weights, max_weight)
    n_frac = node_mask.sum(dim=1).unsqueeze(1) / max_n_nodes
    X_extra = torch.cat((xc, nonlcc.unsqueeze(-1), vec2, cv), dim=-1)
    y_extra = torch.hstack((n_frac, yc, ncomp.unsqueeze(-1), ev5, w))
    return X_extra, y_extra


# --------------------------------------------------------------------------------------
# a13  positional encodings shared between mapped reactant / product atoms
# --------------------------------------------------------------------------------------
def product_positions(atom_map_numbers: Tensor, mol_assignments: Tensor) -> Tensor:
    """neuralnet/transformer_model_with_y.py:400-412 (and :259-275): for every node, the 1-based rank inside the product
    (the molecule with the largest assignment index) of the product atom carrying the node's atom-map number; 0 if none.
    The sequential scatter_ lets the last product atom win on duplicate numbers (0 included)."""
    B, n = atom_map_numbers.shape
    pos = torch.zeros_like(atom_map_numbers)
    for b in range(B):
        pidx = mol_assignments[b].max()
        prod = (mol_assignments[b] == pidx).nonzero().squeeze(1)
        table = torch.zeros(int(atom_map_numbers[b].max()) + 1, dtype=torch.long)
        for k, node in enumerate(prod.tolist(), start=1):
            table[int(atom_map_numbers[b, node])] = k
        pos[b] = table[atom_map_numbers[b]]
    return pos


def sinusoidal_pos_emb(x: Tensor, dim: int) -> Tensor:
    """neuralnet/transformer_model_with_y.py:114-131 (SinusoidalPosEmb)."""
    x = x.reshape(-1).float() * 1000
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half) * -emb).type_as(x)
    emb = x[:, None] * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def pos_enc_sinusoidal(atom_map_numbers: Tensor, mol_assignments: Tensor, dim: int) -> Tensor:
    """:389-417 matched_positional_encodings_sinusoidal, direction 'retro'."""
    B, n = atom_map_numbers.shape
    return sinusoidal_pos_emb(product_positions(atom_map_numbers, mol_assignments), dim).reshape(B, n, dim)


def pos_enc_infoleak(atom_map_numbers: Tensor, dim: int) -> Tensor:
    """:451-490 matched_positional_encodings_infoleak with pos_emb_permutations = 0."""
    B, n = atom_map_numbers.shape
    return sinusoidal_pos_emb(atom_map_numbers, dim).reshape(B, n, dim)


def pos_enc_gaussian(atom_map_numbers: Tensor, noise: Tensor) -> Tensor:
    """:492-503 matched_positional_encodings_gaussian; `noise` (B,n,dim) is the reference's torch.randn draw."""
    noise = noise.clone()
    noise[:, 0, :] = 0
    return torch.stack([noise[b, atom_map_numbers[b]] for b in range(noise.shape[0])])


def pos_enc_laplacian(E_classes: Tensor, atom_map_numbers: Tensor, mol_assignments: Tensor, max_eigenvectors: int, dim: int,
                      signs: Tensor):
    """:221-287 matched_positional_encodings_laplacian (retro) with the dense decomposition of laplacian_eigenvectors_gpu
    (:173-211): product-block graph, unweighted symmetric-normalised Laplacian with unit diagonal (torch_geometric
    get_laplacian(normalization='sym'), :150-154), eigenvectors 1..k of the ascending spectrum times `signs` (B,max_eig),
    rows of the product shared through the atom-map numbers, zero padded to dim.
    Pinned by tests/golden/pos_laplacian.npz: both reference variants run on shims for the three torch_geometric helpers
    they call (tests/golden/make_golden_r2.py), compared up to the sign / basis freedom of eigenvectors
    (tests/golden/common.py check_lap_encodings)."""
    B, n = atom_map_numbers.shape
    out = torch.zeros(B, n, dim)
    pos = product_positions(atom_map_numbers, mol_assignments)
    laps = []
    for b in range(B):
        isprod = mol_assignments[b] == mol_assignments[b].max()
        A = ((E_classes[b] != 0) & isprod[:, None] & isprod[None, :]).float()
        A.fill_diagonal_(0)
        deg = A.sum(-1)
        dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
        Lap = torch.eye(n) - dis[:, None] * A * dis[None, :]
        laps.append(Lap)
        _, vec = torch.linalg.eigh(Lap)
        k = min(max_eigenvectors, int(isprod.sum()))
        pe = vec[:, 1:k + 1] * signs[b, :k]
        table = torch.cat([torch.zeros(1, dim), torch.cat([pe, torch.zeros(n, dim - k)], dim=-1)[isprod]], 0)
        out[b] = table[pos[b]]
    return out, torch.stack(laps)
