"""User API of the reference's diffalign/inference.py on top of the HIP path: `predict_precursors`,
`predict_precursors_from_diffalign`, `predict_with_inpainting`, the model cache and the transition-table cache keyed by
`diffusion_steps` (inference.py:46-107, 188-199, 333-583).

What differs, by necessity: the reference loads a Hydra config and `checkpoints/epoch760.pt` at first use and converts SMILES
with RDKit.  Neither Hydra nor RDKit nor the checkpoint exist next to this package, and chemistry is outside the hot path, so
  * the model and its cfg are registered once with `configure(model=..., cfg=...)` (or built by `load_model`), and
  * SMILES -> dense condition and samples -> result dicts are INJECTED callables (`configure(encoder=..., decoder=...,
    validator=...)`): plug in the reference's own `smiles_to_dense_data` / `_decode_samples` / `Chem.MolFromSmiles` where RDKit
    is installed.  Without a decoder the samples are de-duplicated as dense graphs (exact equality of the collapsed reactant
    side) and returned under a content hash -- no chemistry is invented here.
Everything between those two callables -- duplicate, reverse diffusion, inpainting, the constraint check -- runs on the GPU."""
from __future__ import annotations

import hashlib
import logging
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from .noise_schedule import AbsorbingStateTransitionMaskNoEdge
from .graph import get_index_from_states
from .placeholder import PlaceHolder

log = logging.getLogger(__name__)

_cached_model = None
_cached_cfg = None
_transition_cache: Dict[Tuple[int, int], Any] = {}
_hooks: Dict[str, Optional[Callable]] = {'encoder': None, 'decoder': None, 'validator': None}
_PLACEHOLDER_TYPES = {'none', 'U', 'Au', 'SuNo'}


def configure(model=None, cfg=None, encoder: Optional[Callable] = None, decoder: Optional[Callable] = None,
              validator: Optional[Callable] = None) -> None:
    """Register the (eval-mode) DiscreteDenoisingDiffusionRxn + cfg that `get_model_and_cfg` hands out, and the chemistry hooks:
    encoder(product_smiles, cfg) -> dense PlaceHolder of ONE condition (the reference's smiles_to_dense_data);
    decoder(final_samples, cfg, product_smiles) -> list of result dicts (the reference's _decode_samples);
    validator(smiles) -> bool (the reference's `Chem.MolFromSmiles(s) is not None`)."""
    global _cached_model, _cached_cfg
    if model is not None:
        _cached_model, _cached_cfg = model, (cfg if cfg is not None else model.cfg)
        _transition_cache.clear()
    for k, v in (('encoder', encoder), ('decoder', decoder), ('validator', validator)):
        if v is not None:
            _hooks[k] = v


def load_model(cfg, dataset_infos, checkpoint_file: Optional[str] = None, device='cuda:0'):
    """inference.py:152-186 without the Hydra / dataset-statistics plumbing: build the host class on `device` and load the
    denoiser weights of a reference checkpoint (keys `model.*` of its `model_state_dict`, `module.` prefixes of nn.DataParallel
    stripped as utils/setup.py:300-348 does).  The file is read with weights_only=True."""
    from .diffusion_rxn import DiscreteDenoisingDiffusionRxn
    model = DiscreteDenoisingDiffusionRxn(cfg=cfg, dataset_infos=dataset_infos, device=device)
    if checkpoint_file is not None:
        sd = torch.load(checkpoint_file, map_location='cpu', weights_only=True)
        sd = sd.get('model_state_dict', sd)
        clean = {}
        for k, v in sd.items():
            k = k.replace('module.', '')
            if k.startswith('model.'):
                clean[k[len('model.'):]] = v
        model.model.load_state_dict(clean)
    return model.eval()


def get_model_and_cfg():
    """inference.py:51-73: the cached model and cfg (registered with configure)."""
    if _cached_model is None:
        raise RuntimeError('diffalign_amd.inference: no model registered -- call configure(model=..., cfg=...) once (load_model builds one '
                           'from a cfg, dataset_infos and a reference checkpoint); Hydra and checkpoints/ are not part of this package')
    return _cached_model, _cached_cfg


def _get_or_build_transition_model_eval(model, cfg, diffusion_steps_eval):
    """inference.py:80-107: the eval transition tables for another step count (cheap: a few (T+1,K,K) tables), cached per model."""
    key = (id(model), int(diffusion_steps_eval))
    if key not in _transition_cache:
        node_idx, edge_idx = get_index_from_states(model.dataset_info.atom_decoder, model.dataset_info.bond_decoder,
                                                   list(cfg.diffusion.node_states_to_mask), cfg.diffusion.edge_states_to_mask)
        _transition_cache[key] = AbsorbingStateTransitionMaskNoEdge(
            int(diffusion_steps_eval), model.Xdim_output, model.Edim_output, model.ydim_output, cfg.diffusion.diffuse_edges,
            model.dataset_info.atom_decoder.index('Au'), 0, node_idx, edge_idx, model.device_)
    return _transition_cache[key]


def _ensure_transition_model(model, cfg, diffusion_steps):
    """inference.py:188-199: diffusion_steps must divide T; swap transition_model_eval when it differs from the current one."""
    T = cfg.diffusion.diffusion_steps
    assert T % diffusion_steps == 0, (f'diffusion_steps={diffusion_steps} must divide T={T}. '
                                      f'Valid values: {[d for d in range(1, T + 1) if T % d == 0]}')
    if diffusion_steps != cfg.diffusion.diffusion_steps_eval:
        model.transition_model_eval = _get_or_build_transition_model_eval(model, cfg, diffusion_steps)
        cfg.diffusion.diffusion_steps_eval = diffusion_steps


def smiles_to_dense_data(product_smiles: str, cfg=None, **kwargs) -> PlaceHolder:
    """inference.py:110-149 through the injected encoder (RDKit + graph_builder live outside this package)."""
    if _hooks['encoder'] is None:
        raise RuntimeError('diffalign_amd.inference: no SMILES encoder registered -- configure(encoder=callable(product_smiles, cfg) -> '
                           'dense PlaceHolder), e.g. the reference\'s smiles_to_dense_data where RDKit is installed')
    return _hooks['encoder'](product_smiles, cfg, **kwargs) if kwargs else _hooks['encoder'](product_smiles, cfg)


def _decode_samples_dense(final_samples: PlaceHolder) -> List[Dict[str, Any]]:
    """Fallback of _decode_samples (inference.py:270-330) without chemistry: samples whose collapsed reactant side (nodes and bonds
    up to the supernode) is identical count as one precursor set; score = share of the samples, as the reference scores by
    frequency.  'precursors' is a content hash of that dense graph, NOT a SMILES."""
    X, E = final_samples.X.detach().cpu(), final_samples.E.detach().cpu()
    if X.dim() == 3:
        X, E = X.argmax(-1), E.argmax(-1)
    mol = final_samples.mol_assignments.detach().cpu() if final_samples.mol_assignments is not None else None
    info: Dict[str, Dict[str, int]] = {}
    for k in range(X.shape[0]):
        rct = (mol[k] < mol[k].max()) & final_samples.node_mask[k].cpu() if mol is not None else torch.ones_like(X[k], dtype=torch.bool)
        key = 'graph:' + hashlib.sha1(X[k][rct].numpy().tobytes() + E[k][rct][:, rct].numpy().tobytes()).hexdigest()[:16]
        slot = info.setdefault(key, {'count': 0, 'sample_idx': k})
        slot['count'] += 1
    total = max(1, X.shape[0])
    results = [{'precursors': key, 'score': v['count'] / total, 'sample_data': final_samples.select_by_batch_idx(v['sample_idx']).serialize(),
                'atom_mapping': []} for key, v in info.items()]
    results.sort(key=lambda r: r['score'], reverse=True)
    return results


def _decode_samples(final_samples, cfg, product_smiles=None):
    if _hooks['decoder'] is not None:
        return _hooks['decoder'](final_samples, cfg, product_smiles)
    return _decode_samples_dense(final_samples)


def predict_precursors_from_diffalign(product_smiles: str, n_precursors: int = 1, diffusion_steps: int = 1):
    """inference.py:333-367."""
    model, cfg = get_model_and_cfg()
    _ensure_transition_model(model, cfg, diffusion_steps)
    dense_data = smiles_to_dense_data(product_smiles, cfg).to_device(model.device_)
    with torch.no_grad():          # (the reference: inference_mode; tensors made there cannot be updated outside it -- the static buffers of cached HIP graphs are)
        final_samples = model.sample_for_condition(dense_data=dense_data, n_samples=n_precursors, inpaint_node_idx=None,
                                                   inpaint_edge_idx=None, device=model.device_)
    final_samples.atom_map_numbers, final_samples.mol_assignments = dense_data.atom_map_numbers, dense_data.mol_assignments
    final_samples.node_mask = dense_data.node_mask
    return _decode_samples(final_samples, cfg, product_smiles=product_smiles)[:n_precursors]


def predict_with_inpainting(product_smiles: str, previous_sample_data: dict, inpaint_node_indices: List[int], n_precursors: int = 1,
                            diffusion_steps: int = 1) -> Tuple[List[Dict[str, Any]], Optional[Dict[str, Any]]]:
    """inference.py:370-554: re-generate while the listed nodes (and the bonds among them) stay fixed; a sample counts only if
    every real, non-placeholder atom outside the list changed its type or one of its bonds.  Returns (results, failure_info)."""
    model, cfg = get_model_and_cfg()
    _ensure_transition_model(model, cfg, diffusion_steps)
    dev = model.device_
    dense_data = PlaceHolder.deserialize(previous_sample_data).to_device(dev)
    import torch.nn.functional as F
    orig_X = (dense_data.X[0] if dense_data.X.ndim == 2 else dense_data.X[0].argmax(-1)).long().clone()
    orig_E = (dense_data.E[0] if dense_data.E.ndim == 3 else dense_data.E[0].argmax(-1)).long().clone()
    n = orig_X.shape[0]
    keep = [int(i) for i in inpaint_node_indices]
    # bonds among the kept nodes stay too (inference.py:408-415)
    kt = torch.tensor([i for i in keep if 0 <= i < n], dtype=torch.long, device=dev)
    sub = orig_E[kt][:, kt] if kt.numel() else orig_E[:0, :0]
    ii, jj = torch.triu(sub != 0, 1).nonzero(as_tuple=True)
    inpaint_edge_indices = [[int(kt[a]), int(kt[b])] for a, b in zip(ii.tolist(), jj.tolist())]
    node_mask_row = dense_data.node_mask[0].bool() if dense_data.node_mask is not None else torch.ones(n, dtype=torch.bool, device=dev)
    if dense_data.X.ndim == 2:
        dense_data.X = F.one_hot(dense_data.X.long(), num_classes=model.Xdim_output).float()
    if dense_data.E.ndim == 3:
        dense_data.E = F.one_hot(dense_data.E.long(), num_classes=model.Edim_output).float()
    cond_am, cond_mol, cond_nm = dense_data.atom_map_numbers, dense_data.mol_assignments, dense_data.node_mask
    with torch.no_grad():          # (the reference: inference_mode; tensors made there cannot be updated outside it -- the static buffers of cached HIP graphs are)
        final_samples = model.sample_for_condition(dense_data=dense_data, n_samples=n_precursors, inpaint_node_idx=[keep],
                                                   inpaint_edge_idx=[inpaint_edge_indices] if inpaint_edge_indices else None, device=dev)
    final_samples.atom_map_numbers, final_samples.mol_assignments, final_samples.node_mask = (
        dense_data.atom_map_numbers, dense_data.mol_assignments, dense_data.node_mask)          # duplicated in place by the call
    new_X = final_samples.X.long() if final_samples.X.ndim == 2 else final_samples.X.argmax(-1)
    new_E = final_samples.E.long() if final_samples.E.ndim == 3 else final_samples.E.argmax(-1)
    n_samples_actual = int(new_X.shape[0])
    # atoms the caller asked to change: real, not kept, not a structural placeholder (inference.py:466-483)
    atom_decoder = list(model.dataset_info.atom_decoder)
    placeholder_cls = torch.tensor([atom_decoder.index(a) for a in _PLACEHOLDER_TYPES if a in atom_decoder], device=dev)
    change = node_mask_row.clone()
    if kt.numel():
        change[kt] = False
    change &= ~torch.isin(orig_X, placeholder_cls)
    change_idx = change.nonzero(as_tuple=True)[0]
    if change_idx.numel() == 0:
        log.warning('Inpainting called with no change-atoms; skipping constraint check.')
        filtered = final_samples
    else:
        atom_diff = new_X[:, change_idx] != orig_X[change_idx]
        bond_diff = (new_E[:, change_idx, :] != orig_E[change_idx, :].unsqueeze(0)).any(-1)
        diff = atom_diff | bond_diff                                           # (samples, |change|)
        ok = diff.all(1)
        if int(ok.sum()) == 0:
            named = lambda idx: [{'index': int(p), 'element': atom_decoder[int(orig_X[p])]} for p in idx.tolist()]
            return [], {'n_samples': n_samples_actual, 'requested_change_atoms': named(change_idx),
                        'stuck_atoms': named(change_idx[~diff.any(0)])}
        filtered = final_samples.select_subset(ok)
    return _decode_samples(filtered, cfg, product_smiles=product_smiles)[:n_precursors], None


def predict_precursors(product_smiles: str, n_precursors: int = 1, diffusion_steps: int = 1) -> List[Dict[str, Any]]:
    """inference.py:557-583: [] for an unparsable product; results whose precursors do not all parse are dropped (both checks go
    through the injected validator; without one every string passes)."""
    valid = _hooks['validator'] or (lambda smi: True)
    if not valid(product_smiles):
        return []
    results = predict_precursors_from_diffalign(product_smiles=product_smiles, n_precursors=n_precursors, diffusion_steps=diffusion_steps)
    return [r for r in results if all(valid(s) for s in str(r['precursors']).split('.'))]
