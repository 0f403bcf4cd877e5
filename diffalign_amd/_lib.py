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


class DiffAlignHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DiffAlignHipError(f'{LIB_PATH} is missing: build it with `python -m diffalign_amd.build` '
                                    '(hipcc, gfx950). There is no CPU fallback.')
        _lib = C.CDLL(LIB_PATH)
        _lib.da_last_error.restype = C.c_char_p
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
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def as_u8(mask: torch.Tensor) -> torch.Tensor:
    """bool -> uint8 view (no copy)."""
    m = mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
