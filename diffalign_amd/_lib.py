"""ctypes binding of libdiffalign_hip.so (the C-ABI in include/diffalign_hip.h).

There is NO CPU fallback: every op in this package goes through this library and raises if the library
is missing or the tensors are not on a HIP device."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DA_LIB: developer hook for timing-experiment builds of the same library (tools/build_variant.sh)
LIB_PATH = os.environ.get('DA_LIB') or os.path.join(_HERE, 'libdiffalign_hip.so')
_lib: Optional[C.CDLL] = None

c_f = C.c_void_p  # all device pointers travel as void*
ABI_VERSION = 3      # DA_ABI_VERSION of include/diffalign_hip.h this binding was written against

# Signatures of include/diffalign_hip.h, 'return:arguments' with p = pointer (void*), i = int, q = long long, f = float,
# s = const char*; set on the library once, so that no call site can truncate a size or pass a float as a double
# (tests/test_abi_exports.py checks this table against the header).
SIGNATURES = {
    'da_abi_version': 'i:',
    'da_init': 'i:',
    'da_last_error': 's:',
    'da_mask_dense': 'i:pppiiiip',
    'da_collapse': 'i:pppppiiiip',
    'da_free_masks': 'i:pppiiiiiipippp',
    'da_apply_mask': 'i:ppppppiiiiip',
    'da_to_dense': 'i:ppppppqqiiiippppppp',
    'da_inpaint': 'i:pipippppppiiiip',
    'da_counter_exp_noise': 'i:pqppiiiip',
    'da_posterior_sample': 'i:pppppppppppffppppppppppiiiip',
    'da_sample_discrete': 'i:pppppppppiiiip',
    'da_noise_probs': 'i:pppppppiiiip',
    'da_elbo_rows': 'i:ippppppppppippppfppiiiip',
    'da_masked_sums': 'i:pppppiip',
    'da_ce_grad': 'i:pppppppffppiiiip',
    'da_attn_core_fwd': 'i:ppppppppfppppiiip',
    'da_attn_core_bwd': 'i:pppppppfpppppppppppppqiiip', 'da_attn_core_bwd_workspace': 'q:iii',
    'da_attn_core_fwd_bf16': 'i:ppppppppfppppiiip', 'da_attn_core_bwd_bf16': 'i:pppppppfpppppppppppppqiiip',
    'da_linear_wgrad_mixed': 'i:pipipppqqiip',
    'da_layernorm_bwd': 'i:pppppppqqip',
    'da_res_ln_fwd': 'i:ppipiipppifpppqip', 'da_res_ln_bwd_workspace': 'q:i', 'da_res_ln_bwd': 'i:pppppiipifppipppqqip',
    'da_relu_dropout': 'i:piqpifp', 'da_relu_dropout_bwd': 'i:pppiqfp',
    'da_pool_stats_fwd': 'i:ppppqiiip', 'da_linear_wgrad_workspace': 'q:qii', 'da_linear_wgrad': 'i:pppppqqiip',
    'da_pool_stats_bwd': 'i:pppppiiip',
    'da_vb_grad': 'i:ppppppppppppppffppiiiip',
    'da_adamw': 'i:pppppqfffffifpp', 'da_adamw_devstep': 'i:pppppqfffffpfpp',
    'da_edge_blob_bytes': 'q:iii',
    'da_edge_in_blob_bytes': 'q:ii',
    'da_edge_out_blob_bytes': 'q:ii',
    'da_kv_frag_bytes': 'q:iii',
    'da_x_post_blob_bytes': 'q:ii', 'da_x_in_blob_bytes': 'q:iii', 'da_x_in': 'i:pippppiiip',
    'da_linear': 'i:pipipppiiiiip',
    'da_x_post': 'i:pppppiiip', 'da_x_post16': 'i:pppppiiip', 'da_x_post16_blob_bytes': 'q:ii',
    'da_add_layernorm': 'i:pppppiip',
    'da_x_in_finish': 'i:pppiip',
    'da_edge_in': 'i:pipppiiiip',
    'da_edge_in_scratch_bytes': 'q:iii',
    'da_edge_in_sym': 'i:pippppiiiip',
    'da_edge_in_sym_static': 'i:pipppiiiip',
    'da_edge_stats_table': 'i:ppiiiip',
    'da_edge_stats_skip': 'i:pppiiiip',
    'da_y_in': 'i:ppppppiiiip',
    'da_y_film': 'i:ppppppiiiip',
    'da_qkv_frag': 'i:ppppppiiip',
    'da_kv_pack': 'i:ppipiiip',
    'da_edge_layer': 'i:pppippppppppfiiiiipqp', 'da_edge_layer_table': 'i:pppippppppppfiiiiipqppip', 'da_edge_in_classes': 'i:pipppppiiiip', 'da_edge_in_classes_keep': 'i:pipppppiiiiip',
    'da_edge_layer_stats': 'i:pppippppppppfiiiiipqpipp',
    'da_edge_layer_ex': 'i:pppippppppppfiiiiipqpip' + 'ppi' + 'pippp', 'da_edge_split_scratch_bytes_rows': 'q:qii',
    'da_edge_layer_tuning': 'i:iiiiii',
    'da_edge_split_scratch_bytes': 'q:iii',
    'da_edge_stats': 'i:ppiiiip',
    'da_x_stats': 'i:ppiiiip',
    'da_y_update': 'i:pppppiiiiiiiip', 'da_y_pre': 'i:pppiiiiip', 'da_y_update_film': 'i:ppppppppppiiiiiiip',
    'da_align_index': 'i:pppiip',
    'da_x_out_finish': 'i:pipipppiiip',
    'da_edge_out': 'i:ppipppppiiiiip',
    'da_edge_out_tiles': 'i:ppipppppiiiiip', 'da_edge_out_sample': 'i:ppipppppppppfppqpppiiiiip',
    'da_edge_out_sample_ex': 'i:ppipppppppppfppqpppiiiiiip',
    'da_extra_adjacency': 'i:pppppppiiip',
    'da_extra_cycles': 'i:ppppppiip',
    'da_extra_eigen': 'i:ppqqppppppiip',
    'da_extra_eigen_solve': 'i:ppppppppiip',
    'da_extra_eigen_solve_cached': 'i:ppppppppiipppip',
    'da_extra_eigen_solve_into': 'i:ppppipiiipppip',
    'da_extra_molecular': 'i:pppppfppiiip',
    'da_pos_product_index': 'i:pppppiip',
    'da_pos_sinusoidal': 'i:pppqip',
    'da_pos_gather_rows': 'i:pppiiiip',
    'da_pos_laplacian': 'i:pppiip',
    'da_pos_lap_gather': 'i:pqqpppppiiiip',
}
_CT = {'p': C.c_void_p, 'i': C.c_int, 'q': C.c_longlong, 'f': C.c_float, 's': C.c_char_p}


class DiffAlignHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DiffAlignHipError(f'{LIB_PATH} is missing: build it with `python -m diffalign_amd.build` '
                                    '(hipcc, gfx950). There is no CPU fallback.')
        _lib = C.CDLL(LIB_PATH)
        for name, sig in SIGNATURES.items():
            ret, args = sig.split(':')
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                raise DiffAlignHipError(f'{LIB_PATH} does not export {name}: rebuild it (python -m diffalign_amd.build)') from None
            fn.restype = _CT[ret]
            fn.argtypes = [_CT[a] for a in args]
        if _lib.da_abi_version() != ABI_VERSION:
            v, _lib = _lib.da_abi_version(), None
            raise DiffAlignHipError(f'{LIB_PATH} has ABI version {v}, this binding expects {ABI_VERSION}: rebuild it (python -m diffalign_amd.build)')
        if torch.cuda.is_available():
            # per-device set-up that must not happen under stream capture (da_edge_in_sym_static's scratch address), for the current device;
            # a process that uses several devices calls da_init() once with each of them current
            if _lib.da_init() != 0:
                raise DiffAlignHipError('da_init failed: ' + _lib.da_last_error().decode())
    return _lib


# Tensors handed to ptr() since the last check(): a temporary built inside a call expression (`L.ptr(x.float())`) would
# otherwise be freed as soon as ptr() returns, and the caching allocator could give its block to the NEXT temporary of the
# same expression before the kernel has even been enqueued.  Once the entry point has returned (check()), stream order
# protects the block.
_held = []


def check(rc: int, what: str) -> None:
    _held.clear()
    if rc != 0:
        raise DiffAlignHipError(f'{what} failed (rc={rc}): {lib().da_last_error().decode()}')


def last_error() -> str:
    return lib().da_last_error().decode()


# Bumped by whoever rewrites parameter storage behind torch's back (optim.FlatAdamW.step: one HIP kernel over the flat buffer, which
# torch's per-tensor version counters do not see); the denoiser modules re-pack their bf16 inference weights when it changes.
weights_epoch = 0


def ptr(t: Optional[torch.Tensor], dtype=None) -> C.c_void_p:
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise DiffAlignHipError('diffalign_amd ops need tensors on a HIP device (no CPU fallback); got ' + str(t.device))
    if not t.is_contiguous():
        raise DiffAlignHipError('tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise DiffAlignHipError(f'expected dtype {dtype}, got {t.dtype}')
    _held.append(t)
    return C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    """torch's current stream of the current device as a hipStream_t (the raw-handle query: torch.cuda.current_stream() builds a Stream
    object and resolves the device index in Python, 8 us a call with more than a hundred calls per step)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def as_u8(mask: torch.Tensor) -> torch.Tensor:
    """bool -> uint8 view (no copy)."""
    m = mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
