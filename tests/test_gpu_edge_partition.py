"""GPU parity of the edge-layer kernel's work partition at the BENCHMARK size (B = 64, n = 300: BASELINE.json configs[1]).

At 19,200 query rows the persistent grid (k_edge_layer.hip, EdgePart) walks 9 full rounds of whole rows per XCD chunk, gives
the older wave of every SIMD pair extra half-rounds (`skew`), cuts the 96 remaining rows of each chunk into runs of key tiles
(optionally for the older waves only: `tail_old`) and folds them in k_edge_split_merge; the small parity cases never reach
these branches (B n < 2,048 rows -> no full round).  Here every variant of the partition is run at that size and compared
 (a) with the whole-rows-only walk (no split scratch), bit for bit where the arithmetic is the same, and
 (b) with the fp32 oracle, graph by graph (graphs are independent), on graphs that hold whole rows, skewed half-round rows and
     split rows of different XCD chunks."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_gpu_denoiser import FULL, TINY, build, inputs, rel_l2, KX, KE, D

pytestmark = pytest.mark.gpu


def _tuning(threads=0, skew=-1, tail_old=-1, stagger=0, prio=0, skip_dead=1):
    from diffalign_amd import _lib as L
    L.check(L.lib().da_edge_layer_tuning(threads, skew, tail_old, stagger, prio, skip_dead), 'da_edge_layer_tuning')


@pytest.fixture(autouse=True)
def _restore_tuning():
    yield
    from diffalign_amd import engine
    _tuning(threads=512)
    engine.SPLIT_TAIL = True


def _split_rows(B, n, threads):
    """Rows of the incomplete last round of every XCD chunk (EdgePart with a split scratch)."""
    R, W = B * n, 32 * (threads // 64)
    chunk = (R + 7) // 8
    rows = []
    for x in range(8):
        r0, r1 = x * chunk, min(R, x * chunk + chunk)
        nfull = max(0, r1 - r0) // W
        rows += list(range(r0 + nfull * W, r1))
    return torch.tensor(rows, dtype=torch.long)


def _layer_inputs(B, n, hd, seed):
    from diffalign_amd import synthetic as S
    g = torch.Generator().manual_seed(seed)
    true = S.make_reaction_batch(B, n, seed=seed, P=None)                 # USPTO-like sizes: padded keys and padded query rows
    nm = true['node_mask'].clone()
    nm[::3] = True                                                        # every third graph without padding (the fast path)
    Xh = torch.randn(B, n, hd['dx'], generator=g); yh = torch.randn(B, hd['dy'], generator=g)
    Eh = torch.randn(B, n, n, hd['de'], generator=g)
    Xh, Eh = O.mask_dense(Xh, Eh, nm)
    return Xh, Eh.to(torch.bfloat16), yh, nm


@pytest.mark.parametrize('B,n', [(64, 300), (7, 300), (9, 257)], ids=['bench_B64_n300', 'B7_n300_ragged_chunks', 'B9_n257_partial_tile'])
def test_partition_variants_agree_and_match_oracle(B, n):
    import ctypes as C
    import math
    from diffalign_amd import engine, _lib as L
    from diffalign_amd.engine import _i
    hd = FULL['hidden_dims']
    m, sd = build(FULL, 1, seed=7)
    pm = m._pack(torch.device(D))
    lw, lib, dx, de, R = pm.layers[0], L.lib(), hd['dx'], hd['de'], B * n
    Xh, Eb, yh, nm = _layer_inputs(B, n, hd, seed=5)
    Xd, Ed, yd, nmd = Xh.to(D), Eb.to(D).contiguous(), yh.to(D), L.as_u8(nm.to(D))
    want_stats = engine.edge_stats(Ed, B, n, de)
    fold = lambda t_: torch.stack([t_[:, :, 0].double().sum(1), t_[:, :, 1].double().sum(1), t_[:, :, 2].double().amin(1), t_[:, :, 3].double().amax(1)], 1)
    # what da_edge_layer reads besides E: FiLM vectors, Q and the K/V fragments (engine.layer_main's prologue)
    ye2p1, ebias, yx1, yx2p1 = engine.layer_film(pm, lw, yd, B, engine.film_buffers(pm, B, Xd.device))
    kvf = torch.empty(lib.da_kv_frag_bytes(B, n, dx), dtype=torch.uint8, device=D)
    Qm = torch.empty(R, dx, device=D)
    L.check(lib.da_qkv_frag(L.ptr(Xd), L.ptr(lw['Wqkv']), L.ptr(lw['bqkv']), L.ptr(nmd), L.ptr(Qm), L.ptr(kvf), _i(B), _i(n), _i(dx), L.stream()), 'da_qkv_frag')

    def run(split_on, **tune):
        """da_edge_layer_stats through the C-ABI under one tuning; returns (newX_in, E_out)."""
        _tuning(**tune)
        nsplit = lib.da_edge_split_scratch_bytes(B, n, dx) if split_on else 0
        split = torch.full((nsplit,), 0xFF, dtype=torch.uint8, device=D) if nsplit > 0 else None      # NaN patterns: every slot read was written
        E_out = torch.empty_like(Ed); newX_in = torch.full((R, dx), float('nan'), device=D)
        tail = engine.tail_stats_buffer(pm, B, n, Ed.device).fill_(float('nan'))
        L.check(lib.da_edge_layer_stats(L.ptr(Ed), L.ptr(E_out), L.ptr(Qm), _i(dx), L.ptr(kvf), L.ptr(nmd), L.ptr(ye2p1), L.ptr(ebias),
                                        L.ptr(yx1), L.ptr(yx2p1), L.ptr(newX_in), L.ptr(lw['blob']), C.c_float(1.0 / math.sqrt(dx // pm.n_head)),
                                        _i(B), _i(n), _i(de), _i(dx), _i(pm.dffe), L.ptr(split), C.c_longlong(nsplit), L.ptr(tail),
                                        _i(tail.shape[1]), L.ptr(pm.ticket), L.stream()), 'da_edge_layer_stats')
        torch.cuda.synchronize()
        assert int(pm.ticket.item()) == 0                     # the statistics counter is back at zero
        assert torch.allclose(fold(tail), fold(want_stats), rtol=1e-5, atol=1e-3) and torch.equal(fold(tail)[:, 2:], fold(want_stats)[:, 2:])
        return newX_in, E_out

    base_X, base_E = run(False, threads=512, skew=0, tail_old=0, skip_dead=0)   # whole rows only (rows past the chunk end are skipped), every tile in full
    assert torch.isfinite(base_X).all() and torch.isfinite(base_E.float()).all()
    variants = [dict(threads=512), dict(threads=512, skew=0, tail_old=0), dict(threads=512, skew=1, tail_old=1),
                dict(threads=512, skew=3, tail_old=0), dict(threads=512, skew=2, tail_old=1, stagger=3, prio=1), dict(threads=256), dict(threads=512, skip_dead=0),
                dict(threads=256, skew=0, tail_old=0, skip_dead=0)]
    for tune in variants:
        gX, gE = run(True, **tune)
        assert torch.equal(gE, base_E), tune                             # the E path never depends on the partition
        split = _split_rows(B, n, tune['threads']).to(D)
        whole = torch.ones(R, dtype=torch.bool, device=D); whole[split] = False
        if tune.get('skip_dead', 1) == 0:
            assert torch.equal(gX[whole], base_X[whole]), tune            # same arithmetic on whole rows
        # split rows: the runs' softmax states are folded in another fp32 order; rows with dead tiles (padded keys): count * p
        # instead of p + p + ... in the softmax denominator
        # Padded query rows are "don't care": x_out(newX_in) is masked to zero for them (layers.py:273).  The full walk leaves the
        # uniform average of V there, the dead-tile form leaves yx1 (wV := 0); everything downstream is identical (checked below).
        real_rows = nm.reshape(-1).to(D)
        assert torch.allclose(gX[real_rows], base_X[real_rows], rtol=1e-5, atol=1e-5), (tune, (gX - base_X)[real_rows].abs().max().item())
        assert torch.isfinite(gX).all()
    # (b) the default variant through engine.layer_main against the oracle, graph by graph.  Per XCD chunk of B n / 8 rows the last
    # graph holds the skewed half-round rows and the split rows, the one before it the end of the full rounds.
    _tuning(threads=512)
    engine.SPLIT_TAIL = True
    gX, gE = engine.layer_main(pm, lw, Xd, Ed, yd, nmd, B, n, stats_out=engine.tail_stats_buffer(pm, B, n, Ed.device))
    torch.cuda.synchronize()
    assert torch.equal(gE, base_E)
    per = max(1, B // 8)
    pick = sorted({0, per - 2, per - 1, 3 * per + per - 1, B // 2, B - 2, B - 1} & set(range(B)))
    for b in pick:
        oX, oE, _ = O.xey_layer(sd, 'tf_layers.0', Xh[b:b + 1], Eb[b:b + 1].float(), yh[b:b + 1], nm[b:b + 1], hd['n_head'])
        hX, hE = gX.reshape(B, n, -1)[b].cpu(), gE[b].float().cpu()
        assert rel_l2(hX, oX[0]) <= 2e-2, (b, rel_l2(hX, oX[0]))
        assert rel_l2(hE, oE[0]) <= 2e-2, (b, rel_l2(hE, oE[0]))
        # row by row, so that one misplaced query row cannot hide in a graph-wide norm
        real = nm[b]
        rowX = ((hX - oX[0]).norm(dim=-1) / oX[0].norm(dim=-1).clamp_min(1e-3))[real]
        rowE = ((hE - oE[0]).flatten(1).norm(dim=-1) / oE[0].flatten(1).norm(dim=-1).clamp_min(1e-3))
        assert rowX.max() <= 8e-2, (b, int(rowX.argmax()), rowX.max().item())
        assert rowE.max() <= 5e-2, (b, int(rowE.argmax()), rowE.max().item())


def test_model_at_benchmark_batch_vs_oracle():
    """Whole denoiser (2 layers, full dims) at B = 64, n = 300 -- the batch bench.py runs -- against the oracle on single graphs."""
    B, n, L_ = 64, 300, 2
    m, sd = build(FULL, L_, seed=21)
    hd = FULL['hidden_dims']
    X, E, y, nm, am, pos, mol = inputs(B, n, seed=31, P=None, dx=hd['dx'])
    full = inputs(B, n, seed=32, P=142, dx=hd['dx'])                      # graphs 0-7 and 56-63: 299 real nodes, the bench's own shape
    for t, f in zip((X, E, y, nm, am, pos, mol), full):
        t[:8] = f[:8]; t[56:] = f[56:]
    with torch.no_grad():
        r = m(X.to(D), E.to(D), y.to(D), nm.to(D), am.to(D), pos.to(D), mol.to(D))
    torch.cuda.synchronize()
    assert torch.equal(r[1], r[1].transpose(1, 2))
    for b in (0, 6, 7, 31, 39, 62, 63):
        s_ = slice(b, b + 1)
        o = O.denoiser_forward(sd, X[s_], E[s_], y[s_], nm[s_], L_, hd['n_head'], (KX, KE, 1), am[s_], pos[s_], mol[s_], True)
        for a_, b_, name in ((r[0][s_], o[0], 'X'), (r[1][s_], o[1], 'E'), (r[2][s_], o[2], 'y')):
            assert torch.isfinite(a_).all()
            assert rel_l2(a_, b_) <= 2e-2, (b, name, rel_l2(a_, b_))
        real = nm[b]
        rowE = ((r[1][b].cpu() - o[1][0]).flatten(1).norm(dim=-1) / o[1][0].flatten(1).norm(dim=-1).clamp_min(1e-3))[real]
        assert rowE.max() <= 6e-2, (b, int(rowE.argmax()), rowE.max().item())
