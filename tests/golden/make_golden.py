"""Generate the golden vectors under tests/golden/ by importing the REFERENCE (read-only checkout at
/root/reference) in the build container and running it on seeded synthetic inputs.

Run:  CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" python tests/golden/make_golden.py

Only the produced .npz files travel to the GPU box; the reference itself never does.  Third-party
modules the container lacks (rdkit, torch_geometric, wandb, ...) are registered as MagicMock stubs --
the same trick the reference's own tests/test_noise_schedule.py:33-45 uses -- and the single
third-party function that does arithmetic on this path, torch_geometric.utils.to_dense_batch
(utils/graph.py:549,576), gets a real shim.  torch.Tensor.multinomial is replaced by the
"exponential race" argmax(p / q), q ~ Exp(1), which is bit-identical to torch's CPU multinomial under
the same generator state (asserted below), so draws become a function of explicit noise.
"""
import importlib
import os
import sys
from types import SimpleNamespace
from unittest.mock import MagicMock

os.environ.setdefault('CUDA_VISIBLE_DEVICES', '')
os.environ.setdefault('HIP_VISIBLE_DEVICES', '')
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

for name in ['rdkit', 'rdkit.Chem', 'rdkit.Chem.Draw', 'rdkit.Chem.rdchem', 'rdkit.Chem.rdChemReactions',
             'rdkit.RDLogger', 'seaborn', 'imageio', 'wandb', 'torch_geometric', 'torch_geometric.utils',
             'torch_geometric.utils.sparse', 'torch_geometric.loader', 'torch_geometric.data', 'omegaconf',
             'pytorch_lightning', 'pytorch_lightning.utilities', 'fcd_torch']:
    try:
        importlib.import_module(name)
    except Exception:
        sys.modules[name] = MagicMock()

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from diffalign.neuralnet import layers as ref_layers
from diffalign.neuralnet import transformer_model_with_y as ref_tm
from diffalign.diffusion import noise_schedule as ref_ns
from diffalign.diffusion import diffusion_abstract as ref_da
from diffalign.utils.diffusion import helpers as ref_helpers
from diffalign.utils import graph as ref_graph
from diffalign.utils.placeholder import PlaceHolder as RefPH

from diffalign_amd import synthetic as S
from tests.golden.common import (ATOMS, BONDS, tiny_dims, init_state_dict, counter_noise, NOISE_BASE)


def to_dense_batch(x, batch, *a, **k):
    """pad groups of a sorted batch vector to (B, max_count) + bool mask."""
    B = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=B)
    m = int(counts.max()) if B else 0
    out = x.new_zeros((B, m) + tuple(x.shape[1:]))
    mask = torch.zeros(B, m, dtype=torch.bool)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(batch.numel()) - start[batch]
    out[batch, pos] = x
    mask[batch, pos] = True
    return out, mask


ref_graph.to_dense_batch = to_dense_batch
ref_helpers.to_dense_batch = to_dense_batch

# ---- noise-driven multinomial --------------------------------------------------------------------
_real_multinomial = torch.Tensor.multinomial
_noise_state = {'counter': 0, 'log': []}


def _race_multinomial(self, num_samples, replacement=False, *, generator=None):
    assert num_samples == 1
    q = counter_noise(self.shape, _noise_state['counter'])
    _noise_state['counter'] += 1
    return (self / q).argmax(-1, keepdim=True)


def check_race_equals_multinomial():
    g1 = torch.Generator().manual_seed(123)
    g2 = torch.Generator().manual_seed(123)
    p = torch.rand(4096, 7, generator=torch.Generator().manual_seed(5))
    p[::7, 3] = 0
    a = _real_multinomial(p, 1, generator=g1)
    q = torch.empty_like(p).exponential_(1.0, generator=g2)
    b = (p / q).argmax(-1, keepdim=True)
    assert torch.equal(a, b), 'exponential race != torch.multinomial on this torch build'


def ns(**kw):
    return SimpleNamespace(**kw)


def make_cfg(dims, T=10, T_eval=10, arch='with_y_atommap_number_pos_enc', skip=True, transition='absorbing_masknoedge',
             extra_features=False, loss='ce'):
    return ns(
        diffusion=ns(diffusion_steps=T, diffusion_steps_eval=T_eval, transition=transition,
                     mask_nodes='product_and_sn', mask_edges='product_and_sn',
                     node_states_to_mask=['SuNo', 'none'], edge_states_to_mask=['mol', 'within', 'across'],
                     diffuse_nodes=True, diffuse_edges=True, temperature_scaling_node=1.0,
                     temperature_scaling_edge=1.0, classifier_free_guidance=False,
                     classifier_free_full_unconditioning=False, classifier_free_guidance_weight=0.1,
                     classifier_free_uncond_prob=0.1, classifier_free_drop_within_batch=False,
                     lambda_train=[5, 0], lambda_test=1, ce_lambda=2, denoiser='neuralnet',
                     diffusion_noise_schedule='cosine'),
        neuralnet=ns(architecture=arch, n_layers=dims['n_layers'], hidden_dims=dims['hidden_dims'],
                     hidden_mlp_dims=dims['hidden_mlp_dims'], dropout=0.0, improved=False,
                     extra_features=extra_features, pos_encoding_type='gaussian_pos_enc', num_lap_eig_vectors=20,
                     p_to_r_skip_connection=skip, p_to_r_init=1.0, alignment_type='correct',
                     input_alignment=False, increase_y_dim_for_multigpu=True, use_ema=False, ema_decay=0.999,
                     pos_emb_permutations=0, use_all_gpus=False),
        train=ns(loss=loss, log_every_t=50, log_to_wandb=False, chains_to_save=1),
        test=ns(with_denoising=True, loss_0_repeat=1, n_samples_per_condition=1, elbo_batch_size=None),
        general=ns(log_every_steps=50),
        dataset=ns(nb_dummy_nodes=15),
    )


def make_info():
    return ns(atom_decoder=list(ATOMS), bond_decoder=list(BONDS), valencies=[0] * len(ATOMS),
              atom_weights={a: 1.0 for a in ATOMS}, max_weight=390, bond_orders=[0, 1, 2, 3, 0, 0, 0],
              remove_h=False, nodes_dist=None, max_n_nodes=300,
              node_types=torch.ones(len(ATOMS)), edge_types=torch.ones(len(BONDS)),
              input_dims={'X': len(ATOMS), 'E': len(BONDS), 'y': 1},
              output_dims={'X': len(ATOMS), 'E': len(BONDS), 'y': 0})


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (np_(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


def ph(d, **over):
    dd = dict(d); dd.update(over)
    return RefPH(X=dd['X'].clone(), E=dd['E'].clone(), y=dd['y'].clone(), node_mask=dd['node_mask'].clone(),
                 atom_map_numbers=dd['atom_map_numbers'].clone(), mol_assignments=dd['mol_assignments'].clone())


# ==================================================================================================
def gen_transitions():
    KX, KE = len(ATOMS), len(BONDS)
    nm = torch.tensor([28, 0]); em = torch.tensor([4, 5, 6])
    out = {}
    a10 = ref_ns.AbsorbingStateTransitionMaskNoEdge(timesteps=10, x_classes=KX, e_classes=KE, y_classes=1,
                                                    abs_state_position_x=27, abs_state_position_e=0,
                                                    node_idx_to_mask=nm, edge_idx_to_mask=em)
    gm = torch.Generator().manual_seed(11)
    xm = torch.rand(KX, generator=gm); xm = xm / xm.sum()
    emg = torch.rand(KE, generator=gm); emg = emg / emg.sum()
    m10 = ref_ns.MarginalUniformTransition(x_marginals=xm.clone(), e_marginals=emg.clone(), y_classes=1,
                                           noise_schedule='cosine', timesteps=10, node_idx_to_mask=nm,
                                           edge_idx_to_mask=em)
    u10 = ref_ns.DiscreteUniformTransition(noise_schedule='cosine', timesteps=10, x_classes=KX, e_classes=KE,
                                           y_classes=1, node_idx_to_mask=nm, edge_idx_to_mask=em)
    for tag, tm in (('abs10', a10), ('mar10', m10), ('uni10', u10)):
        for f in ('Qts_x', 'Qts_e', 'Qts_y', 'Qt_bars_x', 'Qt_bars_e', 'Qt_bars_y', 'x_limit', 'e_limit', 'y_limit'):
            out[f'{tag}_{f}'] = getattr(tm, f)
    out['mar_x_marginals'] = xm; out['mar_e_marginals'] = emg
    for T in (100, 500):
        a = ref_ns.AbsorbingStateTransitionMaskNoEdge(timesteps=T, x_classes=KX, e_classes=KE, y_classes=1,
                                                      abs_state_position_x=27, abs_state_position_e=0,
                                                      node_idx_to_mask=nm, edge_idx_to_mask=em)
        sel = [0, 1, 2, T // 2, T - 1, T]
        out[f'abs{T}_sel'] = np.array(sel)
        for f in ('Qts_x', 'Qts_e', 'Qt_bars_x', 'Qt_bars_e'):
            full = getattr(a, f)
            out[f'abs{T}_{f}_sel'] = full[sel]
            out[f'abs{T}_{f}_sum'] = full.double().sum((1, 2))
            out[f'abs{T}_{f}_wsum'] = (full.double() * torch.arange(full[0].numel()).reshape(full[0].shape).double()).sum((1, 2))
    # gather semantic of get_Qt / get_Qt_bar (noise_schedule.py:147-165)
    t = torch.tensor([[3], [7], [10]])
    out['abs10_gather_t'] = t
    out['abs10_gather_Qt_x'] = a10.get_Qt(t, 'cpu').X
    out['abs10_gather_Qtb_e'] = a10.get_Qt_bar(t, 'cpu').E
    save('transitions', **out)


def gen_container_and_masks():
    g = torch.Generator().manual_seed(21)
    B, n = 3, 9
    X = torch.randn(B, n, 5, generator=g); E = torch.randn(B, n, n, 4, generator=g)
    X[0, 2] = 0; E[1, 3, 4] = 0
    node_mask = torch.tensor([[1] * 9, [1] * 6 + [0] * 3, [1] * 4 + [0] * 5]).bool()
    p = RefPH(X=X.clone(), E=E.clone(), y=torch.zeros(B, 1), node_mask=node_mask).mask()
    pc = RefPH(X=X.clone(), E=E.clone(), y=torch.zeros(B, 1), node_mask=node_mask).mask(collapse=True)
    out = dict(in_X=X, in_E=E, node_mask=node_mask, out_X=p.X, out_E=p.E, col_X=pc.X, col_E=pc.E)
    # free-to-diffuse masks on synthetic reactions (ragged: graphs of different sizes in one batch)
    rx = S.make_reaction_batch(5, 40, seed=4)
    rx['X'][4] = rx['X'][3]; rx['E'][4] = rx['E'][3]     # duplicate graph (edge case: identical rows)
    for k in ('node_mask', 'atom_map_numbers', 'mol_assignments'):
        rx[k][4] = rx[k][3]
    full = S.make_reaction_batch(2, 24, seed=5, P=4)     # no padding at all: SuNo rule (i != n-1) bites
    for tag, d in (('rag', rx), ('full', full)):
        for mode in ('product_and_sn', 'atom_mapping', 'sn', 'product'):
            zt = ph(d, X=torch.zeros_like(d['X']), E=torch.zeros_like(d['E']))
            _, mX, mE = ref_graph.apply_mask(orig=ph(d), z_t=zt, atom_decoder=list(ATOMS), bond_decoder=list(BONDS),
                                             mask_nodes=mode, mask_edges=mode, node_states_to_mask=[],
                                             edge_states_to_mask=[], return_masks=True)
            out[f'{tag}_{mode}_mask_X'] = mX[..., 0]
            out[f'{tag}_{mode}_mask_E'] = mE[..., 0]
            assert bool((mX == mX[..., :1]).all()) and bool((mE == mE[..., :1]).all())
        out[f'{tag}_X_idx'] = d['X'].argmax(-1); out[f'{tag}_E_idx'] = d['E'].argmax(-1)
        out[f'{tag}_node_mask'] = d['node_mask']; out[f'{tag}_am'] = d['atom_map_numbers']
        out[f'{tag}_mol'] = d['mol_assignments']
    # apply_mask value semantics incl. as_logits
    d = rx
    g2 = torch.Generator().manual_seed(22)
    zX = torch.randn(d['X'].shape, generator=g2); zE = torch.randn(d['E'].shape, generator=g2)
    for al in (False, True):
        r = ref_graph.apply_mask(orig=ph(d), z_t=ph(d, X=zX, E=zE), atom_decoder=list(ATOMS), bond_decoder=list(BONDS),
                                 mask_nodes='product_and_sn', mask_edges='product_and_sn', node_states_to_mask=[],
                                 edge_states_to_mask=[], as_logits=al)
        out[f'apply_{int(al)}_X'] = r.X; out[f'apply_{int(al)}_E'] = r.E
    out['apply_zX'] = zX; out['apply_zE'] = zE
    save('container_masks', **out)


def gen_layers():
    dims = tiny_dims()
    hd = dims['hidden_dims']
    B, n = 3, 12
    g = torch.Generator().manual_seed(31)
    X = torch.randn(B, n, hd['dx'], generator=g); E = torch.randn(B, n, n, hd['de'], generator=g)
    y = torch.randn(B, hd['dy'], generator=g)
    node_mask = torch.tensor([[1] * 12, [1] * 9 + [0] * 3, [1] * 5 + [0] * 7]).bool()
    Xm, Em = RefPH(X=X.clone(), E=E.clone(), y=y, node_mask=node_mask).mask().X, None
    p = RefPH(X=X.clone(), E=E.clone(), y=y, node_mask=node_mask).mask()
    X, E = p.X, p.E
    layer = ref_layers.XEyTransformerLayer(dx=hd['dx'], de=hd['de'], dy=hd['dy'], n_head=hd['n_head'],
                                           dim_ffX=hd['dim_ffX'], dim_ffE=hd['dim_ffE'], dropout=0.0).eval()
    sd = init_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}, seed=32)
    layer.load_state_dict(sd)
    with torch.no_grad():
        nX, nE, ny = layer.self_attn(X, E, y, node_mask)
        oX, oE, oy = layer(X, E, y, node_mask)
    out = dict(X=X, E=E, y=y, node_mask=node_mask, blk_X=nX, blk_E=nE, blk_y=ny, out_X=oX, out_E=oE, out_y=oy,
               n_head=hd['n_head'])
    for k, v in sd.items():
        out['w::' + k] = v
    save('layer_tiny', **out)


def build_model(arch, dims, skip, in_dims, out_dims):
    if arch == 'with_y':
        return ref_tm.GraphTransformerWithY(n_layers=dims['n_layers'], input_dims=in_dims,
                                            hidden_mlp_dims=dims['hidden_mlp_dims'], hidden_dims=dims['hidden_dims'],
                                            output_dims=out_dims, act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(),
                                            dropout=0.0).eval()
    return ref_tm.GraphTransformerWithYAtomMapPosEmb(n_layers=dims['n_layers'], input_dims=in_dims,
                                                     hidden_mlp_dims=dims['hidden_mlp_dims'],
                                                     hidden_dims=dims['hidden_dims'], output_dims=out_dims,
                                                     act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), dropout=0.0,
                                                     p_to_r_skip_connection=skip, p_to_r_init=1.0,
                                                     alignment_type='correct').eval()


def model_inputs(B, n, seed, P=None, t=5, T=10):
    true = S.make_reaction_batch(B, n, seed=seed, P=P)
    zt = S.noised_state(true, t=t, T=T, abs_x=27, seed=seed + 100)
    g = torch.Generator().manual_seed(seed + 200)
    Xin = torch.cat([zt['X'], torch.rand(B, n, 8, generator=g)], -1)
    yin = torch.cat([zt['y'], torch.rand(B, 12, generator=g)], -1)
    return true, zt, Xin, yin


def gen_models():
    KX, KE = len(ATOMS), len(BONDS)
    in_dims = {'X': KX + 8, 'E': KE, 'y': 13}; out_dims = {'X': KX, 'E': KE, 'y': 1}
    dims = tiny_dims()
    out = {}
    true, zt, Xin, yin = model_inputs(4, 24, seed=41)
    pos = S.shared_pos_encodings(true, dims['hidden_dims']['dx'], seed=43)
    for tag, arch, skip in (('plain', 'with_y', False), ('am_noskip', 'am', False), ('am_skip', 'am', True)):
        m = build_model(arch, dims, skip, in_dims, out_dims)
        sd = init_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=44)
        if skip:
            sd['skip_scaling'] = torch.tensor([0.7]); sd['skip_scaling_2'] = torch.tensor([1.3])
        m.load_state_dict(sd)
        with torch.no_grad():
            if arch == 'with_y':
                r = m(Xin.clone(), zt['E'].clone(), yin.clone(), zt['node_mask'])
            else:
                r = m(Xin.clone(), zt['E'].clone(), yin.clone(), zt['node_mask'], zt['atom_map_numbers'].clone(),
                      pos.clone(), zt['mol_assignments'].clone())
        out[f'{tag}_X'] = r[0]; out[f'{tag}_E'] = r[1]; out[f'{tag}_y'] = r[2]
        if tag == 'am_skip':
            for k, v in sd.items():
                out['w::' + k] = v
    out.update(X_idx=zt['X'].argmax(-1), E_idx=zt['E'].argmax(-1), X_extra=Xin[..., KX:], y_in=yin,
               node_mask=zt['node_mask'], am=zt['atom_map_numbers'], mol=zt['mol_assignments'], pos=pos,
               n_layers=dims['n_layers'], n_head=dims['hidden_dims']['n_head'])
    save('model_tiny', **out)

    # full align_absorbing dims, ONE layer, B=1, n=32 (weights from the seeded initialiser, not stored)
    full = dict(n_layers=1, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128,
                                         'dim_ffy': 128}, hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
    true, zt, Xin, yin = model_inputs(1, 32, seed=51, P=8)
    pos = S.shared_pos_encodings(true, 256, seed=53)
    m = build_model('am', full, True, in_dims, out_dims)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = init_state_dict(shapes, seed=54)
    m.load_state_dict(sd)
    with torch.no_grad():
        r = m(Xin.clone(), zt['E'].clone(), yin.clone(), zt['node_mask'], zt['atom_map_numbers'].clone(), pos.clone(),
              zt['mol_assignments'].clone())
    keys = sorted(shapes)
    save('model_full1', out_X=r[0], out_E=r[1], out_y=r[2], X_idx=zt['X'].argmax(-1), E_idx=zt['E'].argmax(-1),
         X_extra=Xin[..., KX:], y_in=yin, node_mask=zt['node_mask'], am=zt['atom_map_numbers'],
         mol=zt['mol_assignments'], pos=pos, keys=np.array(keys), shapes=np.array([str(shapes[k]) for k in keys]),
         n_params=sum(int(np.prod(s)) for s in shapes.values()))


def gen_posterior_and_draws():
    KX, KE = len(ATOMS), len(BONDS)
    nm = torch.tensor([28, 0]); em = torch.tensor([4, 5, 6])
    out = {}
    for fam in ('abs', 'mar'):
        if fam == 'abs':
            tm = ref_ns.AbsorbingStateTransitionMaskNoEdge(timesteps=10, x_classes=KX, e_classes=KE, y_classes=1,
                                                           abs_state_position_x=27, abs_state_position_e=0,
                                                           node_idx_to_mask=nm, edge_idx_to_mask=em)
        else:
            gm = torch.Generator().manual_seed(11)
            xm = torch.rand(KX, generator=gm); xm = xm / xm.sum()
            emg = torch.rand(KE, generator=gm); emg = emg / emg.sum()
            tm = ref_ns.MarginalUniformTransition(x_marginals=xm, e_marginals=emg, y_classes=1,
                                                  noise_schedule='cosine', timesteps=10, node_idx_to_mask=nm,
                                                  edge_idx_to_mask=em)
        for t in (10, 5, 1):
            true = S.make_reaction_batch(3, 40, seed=61)
            zt = S.noised_state(true, t=t, T=10, abs_x=27, seed=62 + t)
            if fam == 'mar':     # marginal process: scatter a few reactant nodes/edges to random classes
                g0 = torch.Generator().manual_seed(63)
                ridx = torch.randint(1, 26, zt['X'].shape[:2], generator=g0)
                flip = (torch.rand(zt['X'].shape[:2], generator=g0) < 0.3) & (zt['X'].argmax(-1) == 27)
                zt['X'][flip] = F.one_hot(ridx[flip], KX).float()
            g = torch.Generator().manual_seed(64 + t)
            lX = torch.randn(zt['X'].shape, generator=g) * 3; lE = torch.randn(zt['E'].shape, generator=g) * 3
            lX[0, 1] = torch.tensor([-1e4] * KX); lX[0, 1, 2] = 0.    # a hard one-hot prediction
            t_arr = torch.full((3, 1), t, dtype=torch.long)
            pred = ph(zt, X=torch.log_softmax(lX, -1), E=torch.log_softmax(lE, -1))
            prob = ref_helpers.get_p_zs_given_zt(transition_model=tm, t_array=t_arr, pred=pred, z_t=ph(zt),
                                                 return_prob=True)
            _noise_state['counter'] = 1000 * t + (0 if fam == 'abs' else 500)
            c0 = _noise_state['counter']
            torch.Tensor.multinomial = _race_multinomial
            try:
                zs = ref_helpers.sample_discrete_features(prob=prob)
            finally:
                torch.Tensor.multinomial = _real_multinomial
            k = f'{fam}_t{t}_'
            out.update({k + 'logit_X': lX, k + 'logit_E': lE, k + 'zt_X': zt['X'].argmax(-1),
                        k + 'zt_E': zt['E'].argmax(-1), k + 'prob_X': prob.X, k + 'prob_E': prob.E,
                        k + 'zs_X': zs.X.argmax(-1), k + 'zs_E': zs.E.argmax(-1), k + 'noise_counter': c0})
        out[f'{fam}_node_mask'] = zt['node_mask']
    out['true_X'] = true['X'].argmax(-1); out['true_E'] = true['E'].argmax(-1)
    out['am'] = true['atom_map_numbers']; out['mol'] = true['mol_assignments']
    save('posterior_draws', **out)


class TinyDiffusion(ref_da.DiscreteDenoisingDiffusion):
    """Reference host class with injected positional encodings (they are an input at the boundary)."""
    def get_pos_encodings(self, z_t):
        return S.shared_pos_encodings(dict(atom_map_numbers=self._orig_am), self.cfg.neuralnet.hidden_dims['dx'],
                                      seed=73)


def build_diffusion(cfg, seed):
    model = TinyDiffusion(cfg, make_info()).eval()
    shapes = {k: tuple(v.shape) for k, v in model.model.state_dict().items()}
    sd = init_state_dict(shapes, seed=seed)
    model.model.load_state_dict(sd)
    return model, sd


def gen_chain_and_elbo():
    """BASELINE.json configs[0]: tiny 2-layer graph transformer, 10-step absorbing reverse diffusion on 4
    synthetic reaction graphs (CPU reference path), plus ELBO terms and training losses on the same model."""
    dims = tiny_dims()
    cfg = make_cfg(dims, T=10, T_eval=10)
    model, sd = build_diffusion(cfg, seed=71)
    true = S.make_reaction_batch(4, 24, seed=72)
    model._orig_am = true['atom_map_numbers'].clone()
    out = {}
    torch.manual_seed(0)
    torch.Tensor.multinomial = _race_multinomial
    try:
        _noise_state['counter'] = 0
        final = model.sample_one_batch(device='cpu', data=ph(true))
        out['chain_X'] = final.X.argmax(-1); out['chain_E'] = final.E.argmax(-1)
        out['chain_noise_calls'] = _noise_state['counter']
        # one explicit reverse step from a known state, with everything recorded
        zt = S.noised_state(true, t=6, T=10, abs_x=27, seed=74)
        z = ph(zt); z.y = torch.full((4, 1), 6.0)
        with torch.no_grad():
            pred = model.forward(z_t=z, pos_encodings=model.get_pos_encodings(z))
        out['step_logit_X'] = pred.X; out['step_logit_E'] = pred.E
        out['step_zt_X'] = zt['X'].argmax(-1); out['step_zt_E'] = zt['E'].argmax(-1)
        # ELBO
        _noise_state['counter'] = 5000
        with torch.no_grad():
            vb, lt, l0 = model.elbo(ph(true), avg_over_batch=False)
        out['elbo_vb'] = vb; out['elbo_lt'] = lt; out['elbo_l0'] = l0
        out['elbo_noise_calls'] = _noise_state['counter'] - 5000
        with torch.no_grad():
            klp = model.kl_prior(ph(true))
        out['kl_prior_X'] = klp.X; out['kl_prior_E'] = klp.E
        # single compute_Lt with recorded inputs
        t_int = torch.full((4, 1), 4.0)
        _noise_state['counter'] = 6000
        with torch.no_grad():
            z4, _ = model.apply_noise(ph(true), t_int=t_int, transition_model=model.transition_model_eval)
            z4 = ref_graph.apply_mask(orig=ph(true), z_t=z4, atom_decoder=list(ATOMS), bond_decoder=list(BONDS),
                                      mask_nodes='product_and_sn', mask_edges='product_and_sn')
            pred4 = model.forward(z_t=z4, pos_encodings=model.get_pos_encodings(z4))
            Lt = model.compute_Lt(dense_true=ph(true), z_t=z4, t=t_int, x_0_tilde_logit=pred4,
                                  transition_model=model.transition_model_eval)
            Ltl = model.compute_Lt(dense_true=ph(true), z_t=z4, t=t_int, x_0_tilde_logit=pred4,
                                   transition_model=model.transition_model_eval, log=True)
        out.update(lt4_zt_X=z4.X.argmax(-1), lt4_zt_E=z4.E.argmax(-1), lt4_logit_X=pred4.X, lt4_logit_E=pred4.E,
                   lt4_kl_X=Lt.X, lt4_kl_E=Lt.E, lt4log_kl_X=Ltl.X, lt4log_kl_E=Ltl.E)
    finally:
        torch.Tensor.multinomial = _real_multinomial
    # training losses (config 4 shape): ce and vb on a fixed (t, z_t); gradient norms of named params
    for loss in ('ce', 'vb', 'vbce'):
        cfg_l = make_cfg(dims, T=10, T_eval=10, loss=loss)
        m2, _ = build_diffusion(cfg_l, seed=71)
        m2._orig_am = true['atom_map_numbers'].clone()
        m2.train()   # dropout is 0.0 in this config
        t_int = torch.tensor([[3.], [1.], [7.], [10.]])
        torch.Tensor.multinomial = _race_multinomial
        try:
            _noise_state['counter'] = 7000
            z, _ = m2.apply_noise(ph(true), t_int=t_int, transition_model=m2.transition_model)
        finally:
            torch.Tensor.multinomial = _real_multinomial
        z = ref_graph.apply_mask(orig=ph(true), z_t=z, atom_decoder=list(ATOMS), bond_decoder=list(BONDS),
                                 mask_nodes='product_and_sn', mask_edges='product_and_sn')
        z_keep = z.get_new_object()
        pred = m2.forward(z_t=z, pos_encodings=m2.get_pos_encodings(z))
        pred, mX, mE = ref_graph.apply_mask(orig=ph(true), z_t=pred, atom_decoder=list(ATOMS), bond_decoder=list(BONDS),
                                            mask_nodes='product_and_sn', mask_edges='product_and_sn', as_logits=True,
                                            return_masks=True)
        mn = mX.max(-1)[0].flatten(0, -1); me = mE.max(-1)[0].flatten(0, -1)
        if loss == 'ce':
            _, _, L = ref_helpers.ce(pred=pred, discrete_dense_true=ph(true), diffuse_edges=True, diffuse_nodes=True,
                                     lambda_E=5, mask_nodes=mn, mask_edges=me)
        elif loss == 'vb':
            L = m2.elbo_batch_quick(ph(true), pred=pred, z_t=z_keep, lambda_E=5)
        else:
            _, _, cel = ref_helpers.ce(pred=pred, discrete_dense_true=ph(true), diffuse_edges=True, diffuse_nodes=True,
                                       lambda_E=5, mask_nodes=mn, mask_edges=me)
            L = m2.elbo_batch_quick(ph(true), pred=pred, z_t=z_keep, lambda_E=5) + 2 * cel
        L.backward()
        out[f'train_{loss}_loss'] = L.detach()
        for pn in ('mlp_in_E.0.weight', 'tf_layers.0.self_attn.e_out.weight', 'tf_layers.1.linE2.weight',
                   'mlp_out_X.2.weight', 'skip_scaling'):
            gr = dict(m2.model.named_parameters())[pn].grad
            out[f'train_{loss}_gnorm::{pn}'] = gr.norm() if gr is not None else torch.tensor(float('nan'))
        if loss == 'ce':
            out['train_zt_X'] = z_keep.X.argmax(-1); out['train_zt_E'] = z_keep.E.argmax(-1); out['train_t'] = t_int
    for k, v in sd.items():
        out['w::' + k] = v
    out.update(true_X=true['X'].argmax(-1), true_E=true['E'].argmax(-1), node_mask=true['node_mask'],
               am=true['atom_map_numbers'], mol=true['mol_assignments'], n_layers=dims['n_layers'],
               n_head=dims['hidden_dims']['n_head'])
    save('chain_elbo_tiny', **out)


if __name__ == '__main__':
    torch.set_num_threads(8)
    check_race_equals_multinomial()
    which = sys.argv[1:] or ['transitions', 'container', 'layers', 'models', 'posterior', 'chain']
    if 'transitions' in which: gen_transitions()
    if 'container' in which: gen_container_and_masks()
    if 'layers' in which: gen_layers()
    if 'models' in which: gen_models()
    if 'posterior' in which: gen_posterior_and_draws()
    if 'chain' in which: gen_chain_and_elbo()
    print('done')
