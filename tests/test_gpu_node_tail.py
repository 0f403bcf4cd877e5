"""Node tail of a layer (layers.py:266-273, 340-351: x_out + mask, normX1, FFN X, normX2): the two tile shapes of the fused kernel
(da_x_post: 32-row tiles, da_x_post16: 16-row tiles on the 16x16x32 MFMA) against fp32 torch and against each other, at row counts
that are no multiple of a tile or a workgroup, both built widths."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = torch.device('cuda:0')


@pytest.mark.parametrize('dx', [256, 64])
@pytest.mark.parametrize('R', [1, 15, 16, 17, 63, 65, 1000, 7200])
def test_node_tail_tile_shapes_vs_torch(dx, R):
    from diffalign_amd import pack, _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(dx + R)
    rnd = lambda *s: torch.randn(*s, generator=g)
    p = 'tf_layers.0'
    sd = {p + '.self_attn.x_out.weight': rnd(dx, dx) * dx ** -0.5, p + '.self_attn.x_out.bias': rnd(dx) * 0.1,
          p + '.linX1.weight': rnd(dx, dx) * dx ** -0.5, p + '.linX1.bias': rnd(dx) * 0.1,
          p + '.linX2.weight': rnd(dx, dx) * dx ** -0.5, p + '.linX2.bias': rnd(dx) * 0.1,
          p + '.normX1.weight': 1 + 0.1 * rnd(dx), p + '.normX1.bias': 0.1 * rnd(dx),
          p + '.normX2.weight': 1 + 0.1 * rnd(dx), p + '.normX2.bias': 0.1 * rnd(dx)}
    newX_in, X = rnd(R, dx).to(D), rnd(R, dx).to(D)
    mask = (torch.rand(R, generator=g) < 0.8).to(torch.uint8).to(D)
    w = {k[len(p) + 1:]: v.to(D) for k, v in sd.items()}
    nx = F.linear(newX_in, w['self_attn.x_out.weight'], w['self_attn.x_out.bias']) * mask[:, None].float()
    x1 = F.layer_norm(X + nx, (dx,), w['normX1.weight'], w['normX1.bias'])
    ref = F.layer_norm(x1 + F.linear(F.linear(x1, w['linX1.weight'], w['linX1.bias']).relu(), w['linX2.weight'], w['linX2.bias']), (dx,),
                       w['normX2.weight'], w['normX2.bias'])
    outs = []
    for name, blob in (('da_x_post', pack.x_post_blob(sd, p)), ('da_x_post16', pack.x_post16_blob(sd, p))):
        blob = blob.to(D)
        assert blob.numel() == getattr(lib, name + '_blob_bytes')(dx, dx)
        out = torch.full((R + 1, dx), 7.0, device=D)                       # one guard row behind the output
        L.check(getattr(lib, name)(L.ptr(newX_in), L.ptr(X), L.ptr(mask), L.ptr(blob), L.ptr(out), R, dx, dx, L.stream()), name)
        assert bool((out[R] == 7.0).all()), name + ' wrote past its rows'
        err = ((out[:R] - ref).norm() / ref.norm()).item()
        assert err <= 6e-3, (name, err)                                     # bf16 operands, fp32 accumulation and LayerNorms
        outs.append(out[:R])
    assert ((outs[0] - outs[1]).norm() / ref.norm()).item() <= 2e-4         # same roundings of the operands, sums grouped differently
    assert lib.da_x_post16_blob_bytes(128, 128) == -1 and lib.da_x_post16(None, None, None, None, L.ptr(outs[0]), 1, 128, 128, L.stream()) != 0
