"""GPU parity: the HIP denoiser (through the C-ABI) vs the fp32 oracle and the reference's golden logits.

Tolerance (bf16 MFMA operands, bf16 edge activations between layers, fp32 accumulate / LN / softmax):
  rel-L2 <= 2e-2 per output tensor against the fp32 oracle/reference, argmax agreement >= 99 %.
"""
import ast
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import oracle as O
from tests.golden.common import load, init_state_dict, ATOMS, BONDS

pytestmark = pytest.mark.gpu
KX, KE = len(ATOMS), len(BONDS)
T = lambda a: torch.from_numpy(np.asarray(a))
D = 'cuda:0'
FULL = dict(hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
            hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
TINY = dict(hidden_dims={'dx': 64, 'de': 32, 'dy': 32, 'n_head': 4, 'dim_ffX': 64, 'dim_ffE': 64, 'dim_ffy': 64},
            hidden_mlp_dims={'X': 64, 'E': 32, 'y': 32})
IN_DIMS = {'X': KX + 8, 'E': KE, 'y': 13}
OUT_DIMS = {'X': KX, 'E': KE, 'y': 1}


def rel_l2(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(dims, n_layers, skip=True, seed=54):
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
    m = GraphTransformerWithYAtomMapPosEmb(n_layers, IN_DIMS, dims['hidden_mlp_dims'], dims['hidden_dims'], OUT_DIMS,
                                           nn.ReLU(), nn.ReLU(), p_to_r_skip_connection=skip, p_to_r_init=1.0,
                                           alignment_type='correct').eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = init_state_dict(shapes, seed=seed)
    m.load_state_dict(sd)
    return m.to(D), sd


def inputs(B, n, seed, P=None, dx=256):
    from diffalign_amd import synthetic as S
    true = S.make_reaction_batch(B, n, seed=seed, P=P)
    zt = S.noised_state(true, t=5, T=10, abs_x=27, seed=seed + 100)
    g = torch.Generator().manual_seed(seed + 200)
    X = torch.cat([zt['X'], torch.rand(B, n, 8, generator=g)], -1)
    y = torch.cat([zt['y'], torch.rand(B, 12, generator=g)], -1)
    pos = S.shared_pos_encodings(true, dx, seed=seed + 300)
    return X, zt['E'], y, zt['node_mask'], zt['atom_map_numbers'], pos, zt['mol_assignments']


@pytest.mark.parametrize('n', [1, 5, 33, 45, 64, 97])
def test_edge_in_table_path_shapes(n):
    """da_edge_in_sym on symmetric one-hot graphs (table path, float4 and scalar tile loads, partial tiles, all-zero pairs)
    against da_edge_in (MFMA path): bit-identical."""
    from diffalign_amd import engine, _lib as L
    hd = FULL['hidden_dims']
    m, sd = build(FULL, 1, seed=7)
    pm = m._pack(torch.device(D))
    B = 4
    g = torch.Generator().manual_seed(100 + n)
    cl = torch.randint(0, KE, (B, n, n), generator=g)
    cl = torch.triu(cl, 1); cl = cl + cl.transpose(1, 2)
    E = F.one_hot(cl, KE).float()
    nm = torch.rand(B, n, generator=g) < 0.8
    E[3] = E[3] * (nm[3, :, None, None] & nm[3, None, :, None])            # graph 3: masked pairs all zero
    E_d, nm_d = E.to(D), L.as_u8(nm.to(D))
    scratch = torch.empty(int(L.lib().da_edge_in_scratch_bytes(engine._i(B), engine._i(KE), engine._i(hd['de']))), dtype=torch.uint8, device=D)
    a_ = torch.empty(B, n, n, hd['de'], dtype=torch.bfloat16, device=D); b_ = torch.full_like(a_, 7.0)
    args = (engine._i(B), engine._i(n), engine._i(hd['de']), engine._i(FULL['hidden_mlp_dims']['E']), L.stream())
    L.check(L.lib().da_edge_in(L.ptr(E_d), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(a_), *args), 'edge_in')
    L.check(L.lib().da_edge_in_sym(L.ptr(E_d), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(b_), L.ptr(scratch), *args), 'edge_in_sym')
    assert torch.equal(a_, b_)
    assert scratch[:2 * B].cpu().tolist() == [1] * (2 * B)


def test_edge_in_classes_keeps_its_table_between_calls():
    """da_edge_in_classes_keep: a second call on the kept scratch with table_valid = 1 (other input graphs, one of them not one-hot)
    leaves the same class map, flags, class counts and table -- the whole scratch -- and the same hidden rows of the graph that fell
    back, as a fresh da_edge_in_classes call on those graphs."""
    from diffalign_amd import engine, _lib as L
    hd = FULL['hidden_dims']
    m, sd = build(FULL, 1, seed=7)
    pm = m._pack(torch.device(D))
    B, n = 4, 45
    i_ = engine._i

    def graphs(seed, spoil):
        g = torch.Generator().manual_seed(seed)
        cl = torch.randint(0, KE, (B, n, n), generator=g)
        cl = torch.triu(cl, 1); cl = cl + cl.transpose(1, 2)
        E = F.one_hot(cl, KE).float()
        if spoil:
            E[1, 3, 4, 2] = 0.5                                   # graph 1 is not one-hot: it leaves the table path
        return E.to(D), L.as_u8((torch.rand(B, n, generator=g) < 0.8).to(D))

    nbytes = int(L.lib().da_edge_in_scratch_bytes(i_(B), i_(KE), i_(hd['de'])))
    args = (i_(B), i_(n), i_(hd['de']), i_(FULL['hidden_mlp_dims']['E']))
    kept = torch.zeros(nbytes, dtype=torch.uint8, device=D)
    E1, nm1 = graphs(1, False)
    Eh = torch.zeros(B, n, n, hd['de'], dtype=torch.bfloat16, device=D); cm = torch.zeros(B, n, n, dtype=torch.uint8, device=D)
    L.check(L.lib().da_edge_in_classes_keep(L.ptr(E1), i_(KE), L.ptr(nm1), L.ptr(pm.in_E), L.ptr(Eh), L.ptr(cm), L.ptr(kept), *args, i_(0), L.stream()), 'keep 0')
    for seed, spoil in ((2, False), (3, True)):
        E2, nm2 = graphs(seed, spoil)
        Eh_k = torch.zeros_like(Eh); cm_k = torch.zeros_like(cm)
        L.check(L.lib().da_edge_in_classes_keep(L.ptr(E2), i_(KE), L.ptr(nm2), L.ptr(pm.in_E), L.ptr(Eh_k), L.ptr(cm_k), L.ptr(kept), *args, i_(1), L.stream()), 'keep 1')
        fresh = torch.zeros(nbytes, dtype=torch.uint8, device=D)
        Eh_f = torch.zeros_like(Eh); cm_f = torch.zeros_like(cm)
        L.check(L.lib().da_edge_in_classes(L.ptr(E2), i_(KE), L.ptr(nm2), L.ptr(pm.in_E), L.ptr(Eh_f), L.ptr(cm_f), L.ptr(fresh), *args, L.stream()), 'fresh')
        flags = fresh[:2 * B].cpu().tolist()
        assert flags == ([1] * (2 * B) if not spoil else [1, 1, 1, 1, 1, 0, 1, 1]) or (spoil and flags[B + 1] == 0)
        assert torch.equal(kept, fresh)                                      # flags, table, class counts
        ok = torch.tensor([bool(flags[b] and flags[B + b]) for b in range(B)], device=D)
        assert torch.equal(cm_k[ok], cm_f[ok]) and torch.equal(Eh_k[~ok], Eh_f[~ok])


@pytest.mark.parametrize('dims', [FULL, TINY], ids=['full', 'tiny'])
def test_stages_vs_oracle(dims):
    """input stage, one layer, output stage -- each against the oracle on the same (bf16-rounded) inputs."""
    from diffalign_amd import engine, _lib as L
    hd = dims['hidden_dims']
    m, sd = build(dims, 1, seed=7)
    pm = m._pack(torch.device(D))
    B, n = 3, 44
    X, E, y, nm, am, pos, mol = inputs(B, n, seed=11, dx=hd['dx'])
    # -- input stage (edge)
    Eh = torch.empty(B, n, n, hd['de'], dtype=torch.bfloat16, device=D)
    E_d, nm_d, am_d, mol_d = E.to(D), L.as_u8(nm.to(D)), am.to(D), mol.to(D)     # keep device copies alive across the calls
    L.check(L.lib().da_edge_in(L.ptr(E_d), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(Eh), engine._i(B),
                               engine._i(n), engine._i(hd['de']), engine._i(dims['hidden_mlp_dims']['E']), L.stream()), 'edge_in')
    ref = F.relu(F.linear(F.relu(F.linear(E, sd['mlp_in_E.0.weight'], sd['mlp_in_E.0.bias'])), sd['mlp_in_E.2.weight'],
                          sd['mlp_in_E.2.bias']))
    ref = (ref + ref.transpose(1, 2)) / 2
    _, ref = O.mask_dense(torch.zeros(B, n, 1), ref, nm)
    assert rel_l2(Eh.float(), ref) <= 1e-2
    assert torch.equal(Eh.float().cpu()[~(nm[:, :, None] & nm[:, None, :])][:, 0], torch.ones(int((~(nm[:, :, None] & nm[:, None, :])).sum())))
    # -- input stage on a NON-symmetric edge tensor (the mirrored evaluation is skipped only on symmetric tiles)
    Ea = torch.rand(B, n, n, KE, generator=torch.Generator().manual_seed(17))
    Ea_d = Ea.to(D)
    Eh2 = torch.empty_like(Eh)
    L.check(L.lib().da_edge_in(L.ptr(Ea_d), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(Eh2), engine._i(B),
                               engine._i(n), engine._i(hd['de']), engine._i(dims['hidden_mlp_dims']['E']), L.stream()), 'edge_in')
    ref = F.relu(F.linear(F.relu(F.linear(Ea, sd['mlp_in_E.0.weight'], sd['mlp_in_E.0.bias'])), sd['mlp_in_E.2.weight'],
                          sd['mlp_in_E.2.bias']))
    ref = (ref + ref.transpose(1, 2)) / 2
    _, ref = O.mask_dense(torch.zeros(B, n, 1), ref, nm)
    assert rel_l2(Eh2.float(), ref) <= 1e-2
    assert torch.equal(Eh2, Eh2.transpose(1, 2))
    # -- da_edge_in_sym (symmetry established per graph by a coalesced pre-pass): bit-identical, on symmetric, non-symmetric and
    #    mixed batches (graph 1 made non-symmetric in a single entry)
    #    Graphs that are symmetric AND one-hot take the table-gather path, all others the MFMA path: all five mixes below.
    Emix = E.clone(); Emix[1, 3, 5, 2] += 0.25
    Esym = ((Ea + Ea.transpose(1, 2)) / 2); Esym[2] = E[2]                  # symmetric, graphs 0 and 1 not one-hot
    Ehot = E.clone(); Ehot[0, 2, 7] = Ehot[0, 2, 7].roll(1)                   # graph 0 one-hot but not symmetric
    assert not torch.equal(Ehot, E)
    flags = torch.empty(int(L.lib().da_edge_in_scratch_bytes(engine._i(B), engine._i(KE), engine._i(hd['de']))), dtype=torch.uint8, device=D)
    seen = []
    for src_ in (E_d, Ea_d, Emix.to(D), Esym.to(D), Ehot.to(D)):
        a_ = torch.empty_like(Eh); b_ = torch.empty_like(Eh)
        L.check(L.lib().da_edge_in(L.ptr(src_), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(a_), engine._i(B),
                                   engine._i(n), engine._i(hd['de']), engine._i(dims['hidden_mlp_dims']['E']), L.stream()), 'edge_in')
        L.check(L.lib().da_edge_in_sym(L.ptr(src_), engine._i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(b_), L.ptr(flags), engine._i(B),
                                       engine._i(n), engine._i(hd['de']), engine._i(dims['hidden_mlp_dims']['E']), L.stream()), 'edge_in_sym')
        bad = (a_ != b_).any(-1).nonzero()
        assert bad.numel() == 0, (len(seen), flags[:2 * B].cpu().tolist(), bad[:8].tolist(), int(bad.shape[0]))
        # pooled statistics of the stage's output: class counts x table for the graphs the table served, k_e_stats for the rest
        fold = lambda st: torch.stack([st[:, :, 0].sum(1), st[:, :, 1].sum(1), st[:, :, 2].amin(1), st[:, :, 3].amax(1)], 1)
        want, got = fold(engine.edge_stats(b_, B, n, hd['de'])), fold(engine.edge_stats_after_in(flags, b_, B, n, KE, hd['de']))
        assert torch.allclose(got, want, rtol=2e-5, atol=1e-4), (len(seen), (got - want).abs().max().item())
        assert torch.equal(got[:, 2:], want[:, 2:])
        seen.append(flags[:2 * B].cpu().tolist())
    assert seen == [[1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0], [1, 0, 1, 1, 0, 1], [1, 1, 1, 0, 0, 1], [0, 1, 1, 1, 1, 1]], seen
    # -- one layer on random hidden features
    g = torch.Generator().manual_seed(3)
    Xh = torch.randn(B, n, hd['dx'], generator=g); yh = torch.randn(B, hd['dy'], generator=g)
    Ehh = torch.randn(B, n, n, hd['de'], generator=g)
    Xh, Ehh = O.mask_dense(Xh, Ehh, nm)
    Ehh = Ehh.to(torch.bfloat16).float()
    oX, oE, oy = O.xey_layer(sd, 'tf_layers.0', Xh, Ehh, yh, nm, hd['n_head'])
    Eb = Ehh.to(torch.bfloat16).to(D).contiguous()
    st = engine.edge_stats(Eb, B, n, hd['de'])
    Xh_d, yh_d = Xh.to(D).contiguous(), yh.to(D).contiguous()
    gX, gE, gy = engine.layer_forward(pm, pm.layers[0], Xh_d, Eb, yh_d, nm_d, st, B, n)
    torch.cuda.synchronize()
    # the same layer with the Etoy statistics computed by the edge kernel's tail (da_edge_layer_stats) instead of a separate pass
    tail = engine.tail_stats_buffer(pm, B, n, Eb.device).fill_(float('nan'))
    tX, tE = engine.layer_main(pm, pm.layers[0], Xh_d, Eb, yh_d, nm_d, B, n, stats_out=tail)
    assert torch.equal(tX, gX) and torch.equal(tE, gE)
    fold = lambda t_: torch.stack([t_[:, :, 0].double().sum(1), t_[:, :, 1].double().sum(1), t_[:, :, 2].double().amin(1), t_[:, :, 3].double().amax(1)], 1)
    assert torch.allclose(fold(tail), fold(st), rtol=1e-5, atol=1e-4) and torch.equal(fold(tail)[:, 2:], fold(st)[:, 2:])
    _, _, ty = engine.layer_forward(pm, pm.layers[0], Xh_d, Eb, yh_d, nm_d, None, B, n)
    assert torch.allclose(ty, gy, rtol=1e-4, atol=1e-5)
    assert rel_l2(gy, oy) <= 2e-2, rel_l2(gy, oy)
    assert rel_l2(gX, oX) <= 2e-2, rel_l2(gX, oX)
    assert rel_l2(gE.float(), oE) <= 2e-2, rel_l2(gE.float(), oE)
    # -- output stage (edge) incl. skip connection
    from diffalign_amd.engine import _i
    src = torch.empty(B, n, dtype=torch.int32, device=D)
    L.check(L.lib().da_align_index(L.ptr(am_d), L.ptr(mol_d), L.ptr(src), _i(B), _i(n), L.stream()), 'align')
    assert torch.equal(src.cpu().long(), O.align_index(am, mol))
    Eo = torch.empty(B, n, n, KE, device=D)
    s1, s2 = pm.skip
    L.check(L.lib().da_edge_out(L.ptr(Eb), L.ptr(E_d), _i(KE), L.ptr(pm.out_E), L.ptr(src), L.ptr(s1), L.ptr(s2), L.ptr(Eo),
                                _i(B), _i(n), _i(hd['de']), _i(dims['hidden_mlp_dims']['E']), _i(KE), L.stream()), 'edge_out')
    r = F.linear(F.relu(F.linear(Ehh, sd['mlp_out_E.0.weight'], sd['mlp_out_E.0.bias'])), sd['mlp_out_E.2.weight'],
                 sd['mlp_out_E.2.bias']) + E
    r = r * (~torch.eye(n, dtype=torch.bool))[None, :, :, None]
    r = 0.5 * (r + r.transpose(1, 2))
    _, r = O.p_to_r_skip(torch.zeros(B, n, KX), r, X, E, O.align_index(am, mol), 1.0, 1.0, KX, KE)
    assert rel_l2(Eo, r) <= 1e-2, rel_l2(Eo, r)
    assert torch.equal(Eo, Eo.transpose(1, 2))
    # the one-pass tile form (mirrored tiles meet in LDS; what the engine uses): bit-identical, with and without the skip
    # connection, and on a NON-symmetric input edge tensor (the skip term of (j,i) is not the one of (i,j) then)
    for src_, ein_ in ((src, E_d), (None, E_d), (src, Ea_d)):
        a_ = torch.full_like(Eo, 9.0); b_ = torch.full_like(Eo, 7.0)
        sk = (L.ptr(s1), L.ptr(s2)) if src_ is not None else (L.ptr(None), L.ptr(None))
        for fn, dst in ((L.lib().da_edge_out, a_), (L.lib().da_edge_out_tiles, b_)):
            L.check(fn(L.ptr(Eb), L.ptr(ein_), _i(KE), L.ptr(pm.out_E), L.ptr(src_), sk[0], sk[1], L.ptr(dst),
                       _i(B), _i(n), _i(hd['de']), _i(dims['hidden_mlp_dims']['E']), _i(KE), L.stream()), 'edge_out')
        assert torch.equal(a_, b_)


def test_model_full_dims_one_layer_golden():
    """Reference logits (GraphTransformerWithYAtomMapPosEmb, full align_absorbing dims, 1 layer, B=1, n=32)."""
    z = load('model_full1')
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(z['keys'], z['shapes'])}
    m, sd = build(FULL, 1, seed=54)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    X = torch.cat([F.one_hot(T(z['X_idx']), KX).float(), T(z['X_extra'])], -1)
    E = F.one_hot(T(z['E_idx']), KE).float()
    with torch.no_grad():
        r = m(X.to(D), E.to(D), T(z['y_in']).to(D), T(z['node_mask']).to(D), T(z['am']).to(D), T(z['pos']).to(D), T(z['mol']).to(D))
    assert rel_l2(r[0], T(z['out_X'])) <= 2e-2, rel_l2(r[0], T(z['out_X']))
    assert rel_l2(r[1], T(z['out_E'])) <= 2e-2, rel_l2(r[1], T(z['out_E']))
    assert rel_l2(r[2], T(z['out_y'])) <= 2e-2
    agree = (r[1].argmax(-1).cpu() == T(z['out_E']).argmax(-1)).float().mean().item()
    assert agree >= 0.99, agree


@pytest.mark.parametrize('dims,L_,B,n,P', [(FULL, 9, 2, 64, None), (TINY, 2, 4, 24, None), (FULL, 2, 1, 300, 142),
                                           (TINY, 2, 3, 33, 5), (FULL, 1, 1, 97, None), (TINY, 1, 37, 40, None)],
                         ids=['full9', 'tiny2', 'full2_n300', 'tiny_n33_mostly_padding', 'full_n97_B1', 'tiny_B37'])
def test_model_vs_oracle(dims, L_, B, n, P):
    m, sd = build(dims, L_, seed=21)
    hd = dims['hidden_dims']
    X, E, y, nm, am, pos, mol = inputs(B, n, seed=31, P=P, dx=hd['dx'])
    with torch.no_grad():
        r = m(X.to(D), E.to(D), y.to(D), nm.to(D), am.to(D), pos.to(D), mol.to(D))
    o = O.denoiser_forward(sd, X, E, y, nm, L_, hd['n_head'], (KX, KE, 1), am, pos, mol, True)
    for a_, b_, name in ((r[0], o[0], 'X'), (r[1], o[1], 'E'), (r[2], o[2], 'y')):
        assert torch.isfinite(a_).all()
        assert rel_l2(a_, b_) <= 2e-2, (name, rel_l2(a_, b_))
    # argmax agreement on elements whose fp32 top-2 margin is not within the bf16 error band
    def agree(g_, o_, sel):
        top2 = o_.topk(2, -1).values
        clear = sel & ((top2[..., 0] - top2[..., 1]) > 2e-2 * o_.abs().amax(-1).clamp_min(1.0))
        return (g_.argmax(-1).cpu() == o_.argmax(-1))[clear].float().mean().item()
    assert agree(r[0], o[0], nm) >= 0.99
    assert agree(r[1], o[1], nm[:, :, None] & nm[:, None, :]) >= 0.99
    assert torch.equal(r[1], r[1].transpose(1, 2))     # symmetric logits, bit for bit


def test_missing_dims_fail_loudly():
    from diffalign_amd import _lib as L
    odd = dict(hidden_dims={'dx': 96, 'de': 32, 'dy': 16, 'n_head': 4, 'dim_ffX': 64, 'dim_ffE': 64, 'dim_ffy': 64},
               hidden_mlp_dims={'X': 64, 'E': 32, 'y': 32})
    m, _ = build(odd, 1)
    X, E, y, nm, am, pos, mol = inputs(1, 24, seed=1, dx=96)
    with pytest.raises(L.DiffAlignHipError):
        with torch.no_grad():
            m(X.to(D), E.to(D), y.to(D), nm.to(D), am.to(D), pos.to(D), mol.to(D))


@pytest.mark.parametrize('dims', [FULL, TINY], ids=['full', 'tiny'])
def test_x_post_vs_torch(dims):
    """da_x_post (x_out + mask, normX1, FFN, normX2 fused) against the same chain in fp32 torch; rows not a tile multiple."""
    from diffalign_amd import pack, _lib as L
    import ctypes as C
    dx, dff = dims['hidden_dims']['dx'], dims['hidden_dims']['dim_ffX']
    g = torch.Generator().manual_seed(5)
    R = 3 * 41 + 7
    rnd = lambda *s: torch.randn(*s, generator=g)
    sd = {'p.self_attn.x_out.weight': rnd(dx, dx) / math.sqrt(dx), 'p.self_attn.x_out.bias': rnd(dx) * 0.1,
          'p.linX1.weight': rnd(dff, dx) / math.sqrt(dx), 'p.linX1.bias': rnd(dff) * 0.1,
          'p.linX2.weight': rnd(dx, dff) / math.sqrt(dff), 'p.linX2.bias': rnd(dx) * 0.1,
          'p.normX1.weight': 1 + 0.1 * rnd(dx), 'p.normX1.bias': 0.1 * rnd(dx),
          'p.normX2.weight': 1 + 0.1 * rnd(dx), 'p.normX2.bias': 0.1 * rnd(dx)}
    newX_in, X = rnd(R, dx), rnd(R, dx)
    mask = torch.rand(R, generator=g) > 0.2
    newX = F.linear(newX_in, sd['p.self_attn.x_out.weight'], sd['p.self_attn.x_out.bias']) * mask[:, None]
    X1 = F.layer_norm(X + newX, (dx,), sd['p.normX1.weight'], sd['p.normX1.bias'], 1e-5)
    ff = F.linear(F.relu(F.linear(X1, sd['p.linX1.weight'], sd['p.linX1.bias'])), sd['p.linX2.weight'], sd['p.linX2.bias'])
    ref = F.layer_norm(X1 + ff, (dx,), sd['p.normX2.weight'], sd['p.normX2.bias'], 1e-5)
    blob = pack.x_post_blob(sd, 'p').to(D)
    out = torch.full((R, dx), float('nan'), device=D)
    a, b, m = newX_in.to(D), X.to(D), L.as_u8(mask.to(D))
    L.check(L.lib().da_x_post(L.ptr(a), L.ptr(b), L.ptr(m), L.ptr(blob), L.ptr(out), C.c_int(R), C.c_int(dx), C.c_int(dff),
                              L.stream()), 'da_x_post')
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2


@pytest.mark.parametrize('dims,dx_in,with_pos', [(FULL, 37, True), (FULL, 66, False), (TINY, 37, True), (TINY, 29, False), (FULL, 96, True)],
                         ids=['full37', 'full66_nopos', 'tiny37', 'tiny29_nopos', 'full96'])
def test_x_in_vs_torch_and_unfused(dims, dx_in, with_pos):
    """da_x_in (mlp_in_X + hidden mask with encode_no_element + positional encodings, one kernel) against the same chain in fp32 torch
    and against the unfused form (two da_linear + da_x_in_finish); rows not a tile multiple, masked rows, an input width that is
    not a multiple of 16, an all-zero hidden row."""
    from diffalign_amd import pack, _lib as L
    import ctypes as C
    dx, hx = dims['hidden_dims']['dx'], dims['hidden_mlp_dims']['X']
    g = torch.Generator().manual_seed(6)
    R = 5 * 37 + 3
    rnd = lambda *s: torch.randn(*s, generator=g)
    sd = {'mlp_in_X.0.weight': rnd(hx, dx_in) / math.sqrt(dx_in), 'mlp_in_X.0.bias': rnd(hx) * 0.1,
          'mlp_in_X.2.weight': rnd(dx, hx) / math.sqrt(hx), 'mlp_in_X.2.bias': rnd(dx) * 0.1}
    X = rnd(R, dx_in)
    mask = torch.rand(R, generator=g) > 0.2
    mask[0] = True
    pos = rnd(R, dx) if with_pos else None
    hid = F.relu(F.linear(F.relu(F.linear(X, sd['mlp_in_X.0.weight'], sd['mlp_in_X.0.bias'])), sd['mlp_in_X.2.weight'], sd['mlp_in_X.2.bias']))
    ref = hid * mask[:, None]
    ref[ref.sum(-1) == 0, 0] = 1.0
    if pos is not None:
        ref = ref + pos
    assert int(L.lib().da_x_in_blob_bytes(C.c_int(dx_in), C.c_int(hx), C.c_int(dx))) == pack.x_in_blob(sd).numel()
    blob = pack.x_in_blob(sd).to(D)
    out = torch.full((R, dx), float('nan'), device=D)
    Xd, m = X.to(D), L.as_u8(mask.to(D))
    L.check(L.lib().da_x_in(L.ptr(Xd), C.c_int(dx_in), L.ptr(m), L.ptr(pos.to(D)) if with_pos else L.ptr(None), L.ptr(blob), L.ptr(out),
                            C.c_int(R), C.c_int(hx), C.c_int(dx), L.stream()), 'da_x_in')
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and rel_l2(out, ref) < 1e-2, rel_l2(out, ref)
    assert torch.equal(out.cpu()[~mask], ref[~mask])                      # masked rows: exactly e_0 (+ pos)
    # the unfused form on the same bf16 weights: same operand rounding, different summation order
    from diffalign_amd.engine import _linear
    W0, W1 = pack.bf16_weight(sd['mlp_in_X.0.weight']).to(D), pack.bf16_weight(sd['mlp_in_X.2.weight']).to(D)
    h0, un = torch.empty(R, hx, device=D), torch.empty(R, dx, device=D)
    _linear(Xd, dx_in, W0, sd['mlp_in_X.0.bias'].to(D), None, h0, hx, R, hx, dx_in, True)
    _linear(h0, hx, W1, sd['mlp_in_X.2.bias'].to(D), None, un, dx, R, dx, hx, True)
    L.check(L.lib().da_x_in_finish(L.ptr(un), L.ptr(m), L.ptr(pos.to(D)) if with_pos else L.ptr(None), C.c_int(R), C.c_int(dx), L.stream()), 'fin')
    assert (out - un).abs().max() <= 2e-3 * un.abs().max(), float((out - un).abs().max())


def test_x_in_refuses_unbuilt_widths():
    from diffalign_amd import _lib as L
    import ctypes as C
    assert int(L.lib().da_x_in_blob_bytes(C.c_int(97), C.c_int(256), C.c_int(256))) == -1
    assert int(L.lib().da_x_in_blob_bytes(C.c_int(37), C.c_int(96), C.c_int(256))) == -1
    x = torch.zeros(32, 37, device=D); m = torch.ones(32, dtype=torch.uint8, device=D); o = torch.zeros(32, 96, device=D)
    rc = L.lib().da_x_in(L.ptr(x), C.c_int(37), L.ptr(m), L.ptr(None), L.ptr(x), L.ptr(o), C.c_int(32), C.c_int(96), C.c_int(96), L.stream())
    assert rc != 0 and 'not built' in L.last_error()


@pytest.mark.parametrize('dims,L_,B,n,P', [(FULL, 2, 3, 70, None), (TINY, 2, 4, 33, 5), (FULL, 1, 9, 300, 142)], ids=['full_n70', 'tiny_n33', 'full_n300_split_rows'])
def test_first_layer_from_table_equals_materialised_input(dims, L_, B, n, P):
    """da_edge_in_classes + da_edge_layer_table (the first layer takes its input rows from the input stage's table through one class byte
    per pair; the hidden tensor is neither written nor read for graphs that are symmetric and one-hot) against da_edge_in_sym +
    da_edge_layer: the same logits bit for bit -- also when one graph of the batch is NOT one-hot and one is not symmetric (those
    read their materialised rows), and with padded graphs / partial tiles / split rows."""
    from diffalign_amd import engine
    m, sd = build(dims, L_, seed=23)
    hd = dims['hidden_dims']
    X, E, y, nm, am, pos, mol = inputs(B, n, seed=37, P=P, dx=hd['dx'])
    E = E.clone()
    E[1, 2, 3] = 0.5 * (E[1, 2, 3] + E[1, 2, 3].roll(1)); E[1, 3, 2] = E[1, 2, 3]        # graph 1: symmetric, not one-hot
    if B > 2:
        E[2, 4, 5] = E[2, 4, 5].roll(1)                                                    # graph 2: one-hot, not symmetric
    args = [t.to(D) for t in (X, E, y, nm, am, pos, mol)]
    outs = {}
    for flag in (True, False):
        engine.TABLE_IN = flag
        try:
            with torch.no_grad():
                outs[flag] = m(*args)
        finally:
            engine.TABLE_IN = True
    for a_, b_ in zip(outs[True][:3], outs[False][:3]):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_)
    o = O.denoiser_forward(sd, X, E, y, nm, L_, hd['n_head'], (KX, KE, 1), am, pos, mol, True)
    assert rel_l2(outs[True][1], o[1]) <= 2e-2 and rel_l2(outs[True][0], o[0]) <= 2e-2
