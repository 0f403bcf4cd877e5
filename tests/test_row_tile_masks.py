"""CPU: graph.row_tile_masks (the per-row tile masks of da_edge_layer_ex for the last two layers of a fused reverse step) against
a plain loop, including n = 1024 (bit 31: the sign bit of the int32 word) and ragged last tiles."""
import pytest
import torch

from diffalign_amd import graph


@pytest.mark.parametrize('B,n', [(3, 75), (2, 32), (2, 1024), (4, 300)])
def test_row_tile_masks_against_loop(B, n):
    g = torch.Generator().manual_seed(B * 1000 + n)
    mx = torch.rand(B, n, generator=g) < 0.3
    me = torch.zeros(B, n, dtype=torch.bool)
    for b in range(B):                                   # a few runs of free nodes, so that some tiles have none
        for _ in range(2):
            a = int(torch.randint(0, n, (1,), generator=g)); me[b, a:a + int(torch.randint(1, 40, (1,), generator=g))] = True
    if n == 1024:
        me[0, 1000] = True                               # tile 31
    prev, last = graph.row_tile_masks((mx, me))
    assert prev.dtype == last.dtype == torch.int32 and prev.shape == last.shape == (B * n,)
    nt = (n + 31) // 32
    for b in range(B):
        word = 0
        for jt in range(nt):
            if bool(me[b, 32 * jt:32 * jt + 32].any()):
                word |= 1 << jt
        for i in range(0, n, 7):
            want_last = word if bool(me[b, i]) else 0
            assert (int(last[b * n + i]) & 0xffffffff) == want_last
            assert (int(prev[b * n + i]) & 0xffffffff) == (0xffffffff if bool(mx[b, i] | me[b, i]) else 0)
