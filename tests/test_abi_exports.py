"""CPU: the C-ABI library loads and exports every symbol include/diffalign_hip.h declares (no compute calls)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diffalign_amd import _lib as L
    lib = L.lib()
    hdr = open(os.path.join(REPO, 'include', 'diffalign_hip.h')).read()
    names = sorted(set(re.findall(r'\b(da_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 8
    for nme in names:
        assert hasattr(lib, nme), f'{nme} declared in the header but not exported'
    assert lib.da_abi_version() >= 1


def test_signature_table_matches_header():
    """Every declaration of the header has argtypes / restype in diffalign_amd._lib.SIGNATURES, type for type."""
    from diffalign_amd import _lib as L
    hdr = open(os.path.join(REPO, 'include', 'diffalign_hip.h')).read()
    body = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    decls = re.findall(r'^\s*((?:const\s+)?(?:int|long long|char\*|const char\*))\s+(da_\w+)\s*\(([^;]*?)\)\s*;', body, flags=re.M | re.S)

    def code(t):
        t = t.strip()
        if t in ('void', ''):
            return ''
        if '*' in t:
            return 'p'
        for prefix, c in (('long long', 'q'), ('float', 'f'), ('int', 'i')):
            if t.startswith(prefix):
                return c
        raise AssertionError(f'unknown C type {t!r}')

    seen = {}
    for ret, name, args in decls:
        seen[name] = {'int': 'i', 'long long': 'q', 'const char*': 's'}[ret.strip()] + ':' + ''.join(code(a) for a in args.replace('\n', ' ').split(','))
    assert seen == L.SIGNATURES
    lib = L.lib()
    assert lib.da_edge_layer.argtypes[12] is ctypes.c_float and lib.da_kv_frag_bytes.restype is ctypes.c_longlong


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from diffalign_amd import _lib as L
    with pytest.raises(L.DiffAlignHipError):
        L.ptr(torch.zeros(3))


def test_ptr_keeps_temporaries_alive_until_check():
    """A temporary passed to ptr() inside a call expression must survive until the entry point has been called."""
    import weakref
    import torch
    from diffalign_amd import _lib as L
    if not torch.cuda.is_available():
        t = torch.zeros(4)
        L._held.append(t); r = weakref.ref(t); del t
        assert r() is not None
        L.check(0, 'noop')
        assert r() is None
        return
    t = torch.zeros(4, device='cuda')
    r = weakref.ref(t)
    L.ptr(t); del t
    assert r() is not None
    L.check(0, 'noop')
    assert r() is None
