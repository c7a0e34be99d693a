"""CPU: the C-ABI library loads and exports every symbol include/gcn_hip.h declares, and the
ctypes table in gcn_amd/_lib.py has one entry per declaration with the right number of arguments.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declarations():
    text = open(os.path.join(ROOT, 'include', 'gcn_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|const char\*)\s+(gcn_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_and_binding_agree():
    from gcn_amd import _lib
    decls = _declarations()
    assert len(decls) >= 20
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name]) == n, name


def test_library_exports_every_symbol():
    from gcn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declarations():
        assert hasattr(lib, name), name
    loaded = _lib.load()
    assert loaded.gcn_version() >= 100
    assert loaded.gcn_last_error_string() is not None


def test_library_was_built_from_the_sources_on_disk():
    """A stale prebuilt library fails here: the hash compiled into it is the one of the files beside it."""
    from gcn_amd import _lib
    assert len(_lib.build_hash()) == 64
    assert _lib.build_hash() == _lib.source_hash(), 'gcn_amd/libgcn_hip.so is older than its sources: run make -C gcn_amd/csrc'


def test_status_codes_map_to_the_reference_exception_types():
    import pytest
    from gcn_amd import _lib
    _lib.check(_lib.GCN_OK)
    for code in (_lib.GCN_ERR_BAD_ARG, _lib.GCN_ERR_INDEX_RANGE, _lib.GCN_ERR_WORKSPACE):
        with pytest.raises(ValueError):
            _lib.check(code)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.GCN_ERR_HIP)
    # argument validation happens before any HIP call, so it can be exercised without a GPU
    lib = _lib.load()
    assert lib.gcn_spmm_plan_create(-1, 1, 0, None, ctypes.byref(ctypes.c_void_p()), None) == _lib.GCN_ERR_BAD_ARG
    assert b'bad shape' in lib.gcn_last_error_string()
    assert lib.gcn_gemm_f32(4, 4, 4, None, 2, None, 4, 0, None, 4, 0, None) == _lib.GCN_ERR_BAD_ARG
    assert lib.gcn_dropout_f32(4, 4, None, 4, None, 0, None, 4, 0.0, 1, 1, None, 0, None) == _lib.GCN_ERR_BAD_ARG


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: operators reject host tensors instead of computing on the CPU."""
    import pytest
    import torch
    from gcn_amd import sparse
    if torch.cuda.is_available():
        pytest.skip('host-only check')
    indptr = torch.tensor([0, 1], dtype=torch.int64)
    with pytest.raises(ValueError):
        sparse.DeviceCSR(1, 1, indptr, torch.zeros(1, dtype=torch.int32), torch.ones(1))
    with pytest.raises(ValueError):
        sparse.gemm(torch.ones(2, 2), torch.ones(2, 2))
