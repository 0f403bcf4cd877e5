"""GPU: positional encodings on the device (csrc/k_posenc.hip through the C-ABI) against the reference's PositionalEmbedding
outputs (tests/golden/pos_encodings.npz; the Laplacian variants: tests/golden/pos_laplacian.npz, generated from the reference's
code on shims for the three torch_geometric helpers it calls, compared up to the sign / basis freedom of eigenvectors), plus
eigenvector properties and the CPU oracle up to a sign."""
import pytest
import torch

from oracle import oracle as O
from tests.golden.common import load

pytestmark = pytest.mark.gpu
D = torch.device('cuda:0')


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_deterministic_variants_vs_reference_golden(tag):
    from diffalign_amd.pos_encodings import PositionalEmbedding
    z = load('pos_encodings')
    dim = int(z['dim'])
    am, mol = torch.from_numpy(z[f'{tag}::am']).to(D), torch.from_numpy(z[f'{tag}::mol']).to(D)
    pe = PositionalEmbedding(dim, 0)
    pos, prod_node, n_prod = pe.product_positions(am, mol)
    assert torch.equal(pos.cpu().long(), O.product_positions(am.cpu(), mol.cpu()))
    assert torch.equal(n_prod.cpu().long(), (mol == mol.max(-1, keepdim=True).values).sum(-1).cpu())
    # sin / cos of fp32 arguments up to 5e4 (position * 1000): device vs host libm differ by a few ulp of the result
    g = lambda k: torch.from_numpy(z[f'{tag}::{k}'])
    torch.testing.assert_close(pe.matched_positional_encodings_sinusoidal(am, mol).cpu(), g('sinusoidal'), atol=2e-6, rtol=0)
    torch.testing.assert_close(pe.matched_positional_encodings_infoleak(am, mol).cpu(), g('infoleak'), atol=2e-6, rtol=0)
    out = pe.matched_positional_encodings_gaussian(am, mol, noise=g('gaussian_noise').to(D))
    assert torch.equal(out.cpu(), g('gaussian'))


@pytest.mark.parametrize('variant', ['laplacian', 'laplacian_gpu'])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_laplacian_encodings_vs_reference_golden(tag, variant):
    from diffalign_amd.pos_encodings import PositionalEmbedding
    from diffalign_amd import _lib as L
    import ctypes as C
    from tests.golden.common import check_lap_encodings
    z = load('pos_laplacian')
    g = lambda k: torch.from_numpy(z[f'{tag}::{k}'])
    dim, K = int(z['dim']), int(z['K'])
    E, am, mol = g('E').to(D), g('am').to(D), g('mol').to(D)
    B, n = am.shape
    pe = PositionalEmbedding(dim, 0)
    out = getattr(pe, 'matched_positional_encodings_' + variant)(E, am, mol, K, signs=g(f'{variant}_signs').to(D))
    Lk = torch.empty(B, n, n, device=D)
    L.check(L.lib().da_pos_laplacian(L.ptr(E.contiguous()), L.ptr(mol.contiguous()), L.ptr(Lk), C.c_int(B), C.c_int(n), L.stream()), 'lap')
    torch.testing.assert_close(Lk.cpu(), g(f'{variant}_lap'), atol=1e-6, rtol=0)
    done, total = check_lap_encodings(out, g(variant), g(f'{variant}_lap'), am.cpu(), mol.cpu(), K)
    assert done == total and total >= 15, (done, total)


@pytest.mark.parametrize('perms', [3, -1])
def test_infoleak_with_permuted_atom_map_numbers(perms):
    """:458-485: the relabelling uses the host generators as the reference does; seeded alike it gives the reference's encodings."""
    import random
    import numpy as np
    from diffalign_amd.pos_encodings import PositionalEmbedding
    z = load('pos_perm')
    am, mol = torch.from_numpy(z['am']).to(D), torch.from_numpy(z['mol']).to(D)
    np.random.seed(7); random.seed(7)
    out = PositionalEmbedding(int(z['dim']), perms).matched_positional_encodings_infoleak(am, mol)
    torch.testing.assert_close(out.cpu(), torch.from_numpy(z[f'perm{perms}']), atol=2e-6, rtol=0)
    plain = PositionalEmbedding(int(z['dim']), 0).matched_positional_encodings_infoleak(am, mol)
    assert not torch.equal(out, plain) and torch.equal(out[am == 0], plain[am == 0])


def test_laplacian_encodings_properties():
    from diffalign_amd.pos_encodings import PositionalEmbedding
    from diffalign_amd import synthetic as S
    dim, K = 64, 20
    true = S.make_reaction_batch(4, 64, seed=41)
    am, mol, E = true['atom_map_numbers'], true['mol_assignments'], true['E'].argmax(-1)
    signs = (-1 + 2 * torch.randint(0, 2, (4, K), generator=torch.Generator().manual_seed(3))).float()
    pe = PositionalEmbedding(dim, 0)
    out = pe.matched_positional_encodings_laplacian(E.to(D), am.to(D), mol.to(D), K, signs=signs.to(D)).cpu()
    ref, laps = O.pos_enc_laplacian(E, am, mol, K, dim, signs)
    # Laplacian written by the kernel == the oracle's
    Lk = torch.empty(4, 64, 64, device=D)
    from diffalign_amd import _lib as L
    import ctypes as C
    Ed, mold = E.to(D).contiguous(), mol.to(D).contiguous()
    L.check(L.lib().da_pos_laplacian(L.ptr(Ed), L.ptr(mold), L.ptr(Lk), C.c_int(4), C.c_int(64), L.stream()), 'lap')
    torch.testing.assert_close(Lk.cpu(), laps, atol=1e-6, rtol=0)
    for b in range(4):
        isprod = mol[b] == mol[b].max()
        P = int(isprod.sum()); k = min(K, P)
        assert (out[b][:, k:] == 0).all()                                   # zero padding beyond k
        assert (out[b][am[b] == 0] == 0).all()                              # unmapped nodes (dummies, supernode, padding)
        # mapped reactant atom and its product atom share the row
        for r in (mol[b] == 1).nonzero().squeeze(1).tolist():
            p = ((am[b] == am[b, r]) & isprod).nonzero().squeeze(1)
            assert torch.equal(out[b, r], out[b, int(p[0])])
        # every column below the eigenvalue-1 cluster of the isolated nodes is an eigenvector of the product block
        ev = torch.linalg.eigvalsh(laps[b].double())
        V = out[b][isprod][:, :k].double()
        Lp = laps[b][isprod][:, isprod].double()
        for c in range(k):
            lam = float(ev[1 + c])
            if lam < 1 - 1e-3:
                v = V[:, c]
                assert abs(float(v.norm()) - 1) <= 1e-3 and float((Lp @ v - lam * v).abs().max()) <= 2e-3, (b, c)
                if (ev - lam).abs().lt(1e-3).sum() == 1:                     # simple eigenvalue: equals the oracle's up to a sign
                    rv = ref[b][isprod][:, c].double()
                    assert min(float((v - rv).abs().max()), float((v + rv).abs().max())) <= 2e-3, (b, c)


def test_model_uses_builtin_encodings():
    """get_pos_encodings needs no pos_encodings_fn: every pos_encoding_type runs on the device and feeds forward()."""
    from types import SimpleNamespace as ns
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S
    from tests.golden.common import tiny_dims
    from tests.test_gpu_sampling import make_cfg, make_info
    true = S.make_reaction_batch(3, 40, seed=9)
    zt = PlaceHolder(X=true['X'].to(D), E=true['E'].to(D), y=torch.zeros(3, 1, device=D), node_mask=true['node_mask'].to(D),
                     atom_map_numbers=true['atom_map_numbers'].to(D), mol_assignments=true['mol_assignments'].to(D))
    for kind in ('smiles_pos_enc', 'laplacian_pos_enc', 'laplacian_pos_enc_gpu', 'infoleak_pos_enc', 'gaussian_pos_enc'):
        cfg = make_cfg(tiny_dims())
        cfg.neuralnet.pos_encoding_type = kind
        cfg.neuralnet.num_lap_eig_vectors = 20
        model = DiscreteDenoisingDiffusion(cfg, make_info(), device='cuda:0').eval()
        pos = model.get_pos_encodings(zt)
        assert pos.shape == (3, 40, 64) and torch.isfinite(pos).all()
        with torch.no_grad():
            pred = model.forward(zt)
        assert torch.isfinite(pred.X).all() and torch.isfinite(pred.E).all()
