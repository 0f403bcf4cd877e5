"""da_edge_in_sym_static -- the one entry point that allocates nothing (its scratch is a static __device__ array of the library) --
against da_edge_in_sym (caller-owned scratch), through the C-ABI, bit for bit, on two calls in a row, the second smaller."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_denoiser import FULL, TINY, build, KE, D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dims', [FULL, TINY], ids=['full', 'tiny'])
def test_static_scratch_entry_point_matches(dims):
    from diffalign_amd import engine, _lib as L
    from diffalign_amd.engine import _i
    hd, he = dims['hidden_dims'], dims['hidden_mlp_dims']['E']
    m, _ = build(dims, 1, seed=7)
    pm = m._pack(torch.device(D))
    lib = L.lib()
    for call, (B, n) in enumerate([(6, 97), (3, 45)]):                    # second call: fewer, smaller graphs; the array keeps call 1's bytes
        g = torch.Generator().manual_seed(10 + call)
        cl = torch.randint(0, KE, (B, n, n), generator=g)
        cl = torch.triu(cl, 1); cl = cl + cl.transpose(1, 2)
        E = F.one_hot(cl, KE).float()
        E[1, 2, 3, :] = torch.rand(KE, generator=g)                        # graph 1: neither symmetric nor one-hot -> MFMA path
        nm = torch.rand(B, n, generator=g) < 0.8
        E_d, nm_d = E.to(D), L.as_u8(nm.to(D))
        scratch = torch.empty(int(lib.da_edge_in_scratch_bytes(B, KE, hd['de'])), dtype=torch.uint8, device=D)
        want = torch.full((B, n, n, hd['de']), 7.0, dtype=torch.bfloat16, device=D)
        got = torch.full_like(want, 9.0)
        L.check(lib.da_edge_in_sym(L.ptr(E_d), _i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(want), L.ptr(scratch), _i(B), _i(n), _i(hd['de']),
                                   _i(he), L.stream()), 'da_edge_in_sym')
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        L.check(lib.da_edge_in_sym_static(L.ptr(E_d), _i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(got), _i(B), _i(n), _i(hd['de']), _i(he),
                                          L.stream()), 'da_edge_in_sym_static')
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.equal(got, want), (call, int((got != want).sum()))
        assert scratch[:2 * B].cpu().tolist() == [1, 0] + [1] * (B - 2) + [1, 0] + [1] * (B - 2)
    # under stream capture (what the entry point is for): the scratch address was resolved by da_init() when the library was loaded, so the
    # captured call only enqueues memsets and kernels; the replayed graph writes the same bits
    cap = torch.full_like(want, 5.0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            L.check(lib.da_edge_in_sym_static(L.ptr(E_d), _i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(cap), _i(B), _i(n), _i(hd['de']), _i(he),
                                              L.stream()), 'da_edge_in_sym_static (captured)')
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        cap.fill_(5.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, want)
    # sizes beyond the static array are refused with a clear error, nothing is launched
    rc = lib.da_edge_in_sym_static(L.ptr(E_d), _i(KE), L.ptr(nm_d), L.ptr(pm.in_E), L.ptr(got), _i(4097), _i(n), _i(hd['de']), _i(he), L.stream())
    assert rc != 0 and b'4096' in lib.da_last_error()
    L._held.clear()
