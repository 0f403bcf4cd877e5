"""Denoiser modules with the reference's nn.Module contract (diffalign/neuralnet/transformer_model_with_y.py:
GraphTransformerWithY :33-112, GraphTransformerWithYAtomMapPosEmb :557-714): same constructor arguments, same
state-dict keys/shapes, same forward signature and 4-tuple result.  forward() runs in libdiffalign_hip.so
(bf16 MFMA, fp32 accumulate) -- no eager fallback.  In train() mode with autograd enabled the modules run the differentiable
fp32 training forward of train_forward.py instead (HIP autograd.Functions + library GEMMs)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import engine
from .layers import XEyTransformerLayer


class _DenoiserBase(nn.Module):
    def _build(self, n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, act_fn_in, act_fn_out, improved,
               dropout):
        self.n_layers = n_layers
        self.improved = bool(improved)       # NodeEdgeBlockImproved has no fused inference kernels: forward() always takes train_forward's path
        self.n_head = hidden_dims['n_head']
        self.out_dim_X, self.out_dim_E, self.out_dim_y = output_dims['X'], output_dims['E'], output_dims['y']
        if not isinstance(act_fn_in, nn.ReLU) or not isinstance(act_fn_out, nn.ReLU):
            raise NotImplementedError('only ReLU activations are built (the reference never passes anything else)')
        mk = lambda a, b, c, last_act: nn.Sequential(*([nn.Linear(a, b), nn.ReLU(), nn.Linear(b, c)] + ([nn.ReLU()] if last_act else [])))
        self.mlp_in_X = mk(input_dims['X'], hidden_mlp_dims['X'], hidden_dims['dx'], True)
        self.mlp_in_E = mk(input_dims['E'], hidden_mlp_dims['E'], hidden_dims['de'], True)
        self.mlp_in_y = mk(input_dims['y'], hidden_mlp_dims['y'], hidden_dims['dy'], True)
        self.tf_layers = nn.ModuleList([XEyTransformerLayer(dx=hidden_dims['dx'], de=hidden_dims['de'], dy=hidden_dims['dy'],
                                                            n_head=hidden_dims['n_head'], dim_ffX=hidden_dims['dim_ffX'],
                                                            dim_ffE=hidden_dims['dim_ffE'], improved=improved, dropout=dropout)
                                        for _ in range(n_layers)])
        self.mlp_out_X = mk(hidden_dims['dx'], hidden_mlp_dims['X'], output_dims['X'], False)
        self.mlp_out_E = mk(hidden_dims['de'], hidden_mlp_dims['E'], output_dims['E'], False)
        self.mlp_out_y = mk(hidden_dims['dy'], hidden_mlp_dims['y'], output_dims['y'], False)   # unused by forward, kept for checkpoints
        self._packed = None
        self._packed_key = None
        self._plist = None      # list(self.parameters()), dropped when the module tree is rebuilt by _apply (.to(), .float(), ...)

    def _pack(self, device):
        # re-pack when any parameter was rewritten (torch's version counters), moved (storage pointers) or stepped behind torch's back
        # (weights_epoch).  The parameter list is kept: walking the module tree costs more than the whole comparison (531 tensors).
        if self._plist is None:
            self._plist = list(self.parameters())
        key = (str(device), L.weights_epoch, [p.data_ptr() for p in self._plist], [p._version for p in self._plist])
        if self._packed is None or self._packed_key != key:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._packed = engine.PackedModel(sd, device, self.n_layers, self.n_head)
            self._packed_key = key
        return self._packed

    def _apply(self, fn, *a, **k):
        self._plist = None
        return super()._apply(fn, *a, **k)

    # 'bf16' (default): the fused inference path (engine.model_forward: bf16 MFMA operands, fp32 accumulation, edge activations
    # stored bf16 between layers).  'fp32': inference through the training forward's kernels with every tensor and GEMM in fp32
    # (train_forward.py under no_grad; eval() switches its dropout off) -- several times slower and memory-hungry, there to show a
    # chain on the HIP path without bf16 near-tie flips (tests/test_gpu_sampling.py checks configs[0] against the reference
    # draw for draw) and to separate rounding from logic when a checkpoint disagrees.
    inference_precision = 'bf16'

    def _differentiable(self):
        """train() mode with autograd on: the fp32 training forward (train_forward.py) instead of the fused bf16 inference path."""
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _unfused(self):
        if self.inference_precision not in ('bf16', 'fp32'):
            raise ValueError(f"inference_precision {self.inference_precision!r}: 'bf16' or 'fp32'")
        return self._differentiable() or self.improved or self.inference_precision == 'fp32'

    def _unfused_precision(self):
        return 'fp32' if self.inference_precision == 'fp32' and not self._differentiable() else None

    def supports_fused_step(self):
        """da_edge_out_sample (the reverse step's edge half inside the output kernel) is built for 7 edge classes."""
        return self.out_dim_E == 7 and not self.improved and self.inference_precision == 'bf16'

    def _check(self, X):
        if not X.is_cuda:
            raise L.DiffAlignHipError('denoiser inputs must be on a HIP device (no CPU fallback)')


class GraphTransformerWithY(_DenoiserBase):
    def __init__(self, n_layers: int, input_dims: dict, hidden_mlp_dims: dict, hidden_dims: dict, output_dims: dict,
                 act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), improved=False, dropout=0.1):
        super().__init__()
        self._build(n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, act_fn_in, act_fn_out, improved, dropout)

    def forward(self, X, E, y, node_mask, step=None):
        self._check(X)
        if self._unfused():
            _join_inputs(step)
            from .train_forward import forward_train
            return forward_train(self, X, E, y, node_mask, precision=self._unfused_precision()) + (node_mask,)
        pm = self._pack(X.device)
        Xo, Eo, yh = engine.model_forward(pm, X, E, y, node_mask, step=step)
        yo = None if yh is None else yh[..., :self.out_dim_y] + y[..., :self.out_dim_y]    # (None: fused step with a row list)
        return Xo, Eo, yo, node_mask


def _join_inputs(step):
    """The caller's stream waits for inputs still being written on another one (compute_extra_data: the eigensolver's columns of X
    and y) -- for the paths that read X or y before engine.model_forward, which otherwise places that wait itself."""
    pending = step.pop('inputs_ready', None) if step is not None else None
    if pending is not None:
        torch.cuda.current_stream().wait_event(pending[0])


class GraphTransformerWithYAtomMapPosEmb(_DenoiserBase):
    def __init__(self, n_layers: int, input_dims: dict, hidden_mlp_dims: dict, hidden_dims: dict, output_dims: dict,
                 act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), pos_emb_permutations: int = 0, improved=False, dropout=0.1,
                 p_to_r_skip_connection=False, p_to_r_init=10., alignment_type='old', input_alignment=False):
        super().__init__()
        if input_alignment:                                              # :575-580: the aligned product features widen both inputs
            input_dims = dict(input_dims)
            input_dims['X'] += output_dims['X']
            input_dims['E'] += output_dims['E']
        if (p_to_r_skip_connection or input_alignment) and alignment_type not in ('old', 'correct'):
            raise ValueError(f'Alignment type not set correctly ({alignment_type}). Should be old or correct')       # :554
        self.pos_emb_permutations = pos_emb_permutations
        self.p_to_r_skip_connection = p_to_r_skip_connection
        self.alignment_type = alignment_type
        self.input_alignment = input_alignment
        self._build(n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, act_fn_in, act_fn_out, improved, dropout)
        if p_to_r_skip_connection:
            self.skip_scaling = nn.Parameter(torch.tensor([p_to_r_init], dtype=torch.float))
            self.skip_scaling_2 = nn.Parameter(torch.tensor([p_to_r_init], dtype=torch.float))
            self.skip_scaling_3 = nn.Parameter(torch.tensor([1.], dtype=torch.float))

    def forward(self, X, E, y, node_mask, atom_map_numbers, pos_encodings, mol_assignments, step=None):
        assert atom_map_numbers is not None
        self._check(X)
        if self._unfused() or self.input_alignment:
            _join_inputs(step)
        if self._unfused():
            from .train_forward import forward_train
            return forward_train(self, X, E, y, node_mask, atom_map_numbers, pos_encodings, mol_assignments,
                                 precision=self._unfused_precision()) + (node_mask,)
        pm = self._pack(X.device)
        if self.input_alignment:                                         # :630-647
            from .train_forward import aligned_product_features
            X_al, E_al, _ = aligned_product_features(X.float(), E.float(), atom_map_numbers, mol_assignments, self.out_dim_X, self.out_dim_E,
                                                     self.alignment_type)
            X, E = torch.cat([X.float(), X_al], -1), torch.cat([E.float(), E_al], -1)
        src = None
        if self.p_to_r_skip_connection and self.alignment_type == 'old':     # rank alignment (:526-538) instead of da_align_index
            from .train_forward import align_index
            src = align_index(atom_map_numbers, mol_assignments, 'old')
        Xo, Eo, yh = engine.model_forward(pm, X, E, y, node_mask, atom_map_numbers, pos_encodings, mol_assignments,
                                          use_skip=self.p_to_r_skip_connection, step=step,
                                          edge_skip_uniform=self.input_alignment, align_src=src)
        yo = None if yh is None else yh[..., :self.out_dim_y] + y[..., :self.out_dim_y]    # (None: fused step with a row list)
        return Xo, Eo, yo, node_mask
