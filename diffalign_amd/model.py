"""`DiffAlignModel` with the reference's constructor and methods (diffalign/model.py:26-141): the Syntheseus adapter over
`sample_for_condition`.  Syntheseus is optional here: when it is importable the class derives from its
`ExternalBackwardReactionModel` and returns `SingleProductReaction`s through Syntheseus' own helper, exactly like the reference;
when it is not, the same class stands on a minimal base (model_dir, device, default_num_results) and `_get_reactions` returns,
per input, a list of (reactants, {'probability': p}) pairs in the same order."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Sequence

import torch

from . import inference

try:                                                     # pragma: no cover - syntheseus is not in this image
    from syntheseus.reaction_prediction.inference.base import ExternalBackwardReactionModel as _Base
    from syntheseus.reaction_prediction.utils.inference import process_raw_smiles_outputs_backwards as _to_reactions
except Exception:
    _to_reactions = None

    class _Base:
        """What DiffAlignModel uses of syntheseus' ExternalBackwardReactionModel."""

        def __init__(self, model_dir=None, device=None, remove_duplicates=True, use_cache=False, default_num_results=100, **kwargs):
            self.model_dir = Path(model_dir) if model_dir is not None else self.get_default_model_dir()
            self.device = device if device is not None else ('cuda:0' if torch.cuda.is_available() else 'cpu')
            self.remove_duplicates, self.use_cache, self.default_num_results = remove_duplicates, use_cache, default_num_results

        def __call__(self, inputs, num_results=None):
            return self._get_reactions(list(inputs), num_results or self.default_num_results)


class DiffAlignModel(_Base):
    """model.py:45-141.  diffusion_steps must divide the training steps; samples_per_product = minimum number of chains per
    product (the model scores precursor sets by their frequency among the samples)."""

    def __init__(self, *args, diffusion_steps: int = 1, samples_per_product: int = 100, **kwargs):
        self.diffusion_steps = diffusion_steps
        self.samples_per_product = samples_per_product
        super().__init__(*args, **kwargs)
        self._model, self._cfg = inference.get_model_and_cfg()
        self._transition_cache: Dict[int, object] = {}

    @property
    def name(self) -> str:
        return 'DiffAlign'

    def get_default_model_dir(self) -> Path:
        return Path(__file__).resolve().parents[1] / 'checkpoints'

    def get_parameters(self):
        return self._model.parameters()

    def _get_reactions(self, inputs: List, num_results: int) -> List[Sequence]:
        """model.py:67-138: per product, max(num_results, samples_per_product) chains, reactant sets ranked by frequency."""
        results = []
        n_samples = max(num_results, self.samples_per_product)
        cfg = self._cfg
        for mol in inputs:
            product_smiles = getattr(mol, 'smiles', mol)
            dense_data = inference.smiles_to_dense_data(product_smiles, cfg).to_device(self._model.device_)
            inference._ensure_transition_model(self._model, cfg, self.diffusion_steps)
            with torch.no_grad():          # (the reference: inference_mode; tensors made there cannot be updated outside it -- the static buffers of cached HIP graphs are)
                final_samples = self._model.sample_for_condition(dense_data=dense_data, n_samples=n_samples, inpaint_node_idx=None,
                                                                 inpaint_edge_idx=None, device=self._model.device_)
            final_samples.atom_map_numbers, final_samples.mol_assignments, final_samples.node_mask = (
                dense_data.atom_map_numbers, dense_data.mol_assignments, dense_data.node_mask)
            decoded = inference._decode_samples(final_samples, cfg, product_smiles=product_smiles)
            top = decoded[:num_results]
            outputs = [r['precursors'] for r in top]
            meta = [{'probability': r['score']} for r in top]
            if _to_reactions is not None:
                results.append(_to_reactions(input=mol, output_list=outputs, metadata_list=meta))
            else:
                results.append(list(zip(outputs, meta)))
        return results
