"""ctypes binding of ``libgcn_hip.so`` (declarations mirror ``include/gcn_hip.h``).

The library is the only compute path of this package: if it is missing, or a
call fails, an exception is raised -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgcn_hip.so')

GCN_OK = 0
GCN_ERR_BAD_ARG = 1
GCN_ERR_INDEX_RANGE = 2
GCN_ERR_HIP = 3
GCN_ERR_WORKSPACE = 4
GCN_ERR_RCCL = 5

GCN_ADD_SELF_LOOPS = 1
GCN_BINARY = 2
GCN_RMAT_NATURAL_ORDER = 8
GCN_RELU = 1
GCN_SPLIT_WIDE = 4
GCN_NORM_SYM = 0
GCN_NORM_RW = 1

_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_int = ctypes.c_int
_f32 = ctypes.c_float
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_pi64 = ctypes.POINTER(ctypes.c_int64)
_psz = ctypes.POINTER(ctypes.c_size_t)

# name -> argtypes, in header order.  tests/test_abi.py checks this table against the header.
SIGNATURES = {
    'gcn_version': [],
    'gcn_last_error_string': [],
    'gcn_build_hash': [],
    'gcn_coo_to_csr_bound': [_i64, _i64, _int, _pi64, _psz],
    'gcn_coo_to_csr': [_i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_rows_to_indptr': [_i64, _i64, _vp, _vp, _vp],
    'gcn_row_sums': [_i64, _vp, _vp, _vp, _vp],
    'gcn_degree_scale': [_i64, _vp, _int, _vp, _vp],
    'gcn_normalize_adj': [_i64, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp],
    'gcn_csr_transpose_bound': [_i64, _i64, _i64, _psz],
    'gcn_csr_transpose': [_i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_degree_order_bound': [_i64, _psz],
    'gcn_degree_order': [_i64, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_hub_order_bound': [_i64, _psz],
    'gcn_hub_order': [_i64, _vp, _i64, _vp, _vp, _vp, _sz, _vp],
    'gcn_csr_relabel_bound': [_i64, _i64, _int, _psz],
    'gcn_csr_relabel': [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_check_indices': [_i64, _vp, _i64, _vp],
    'gcn_rmat_csr_bound': [_int, _int, _int, _pi64, _psz],
    'gcn_rmat_csr': [_int, _int, _u64, _int, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_rmat_rows_count': [_int, _int, _u64, _int, _i64, _i64, _pi64, _vp],
    'gcn_rmat_csr_rows_bound': [_i64, _psz],
    'gcn_rmat_csr_rows': [_int, _int, _u64, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp],
    'gcn_fill_synthetic_f32': [_i64, _u64, _vp, _vp],
    'gcn_fill_synthetic_range_f32': [_i64, _i64, _u64, _vp, _vp],
    'gcn_spmm_plan_create': [_i64, _i64, _i64, _vp, ctypes.POINTER(_vp), _vp],
    'gcn_spmm_plan_create_split': [_i64, _i64, _i64, _vp, _i64, ctypes.POINTER(_vp), _vp],
    'gcn_spmm_plan_destroy': [_vp],
    'gcn_spmm_plan_set_hot_columns': [_vp, _i64],
    'gcn_spmm_plan_set_lds_hub_rows': [_vp, _i64],
    'gcn_spmm_plan_info': [_vp, _pi64],
    'gcn_spmm_plan_workspace_bytes': [_vp, _i64, _psz],
    'gcn_spmm_plan_set_timing': [_vp, _int],
    'gcn_spmm_plan_last_timing': [_vp, ctypes.POINTER(ctypes.c_float)],
    'gcn_spmm_csr_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _sz, _vp],
    'gcn_spmm_csr_f32_noalloc': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp],
    'gcn_spmm_csr_two_part_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _int, _f32, _u64, _u64,
                                  _vp, _i64, _i64, _vp, _sz, _vp],
    'gcn_spmm_csr_add_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _sz, _vp],
    'gcn_pack_pitch': [_i64, _pi64],
    'gcn_pack_rows_f32': [_i64, _i64, _vp, _i64, _vp, _i64, _vp],
    'gcn_dropout_pack_f32': [_i64, _i64, _vp, _i64, _f32, _u64, _u64, _vp, _i64, _i64, _vp, _i64, _vp],
    'gcn_dropout_pack_rows_f32': [_i64, _i64, _vp, _i64, _f32, _u64, _u64, _vp, _i64, _i64, _vp, _vp, _i64, _vp],
    'gcn_gemm_gated_pack_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp],
    'gcn_gemm_gated_pack_rows_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _i64, _vp],
    'gcn_unpack_rows_f32': [_i64, _i64, _vp, _i64, _vp, _i64, _vp],
    'gcn_spmm_csr_packed_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _sz, _vp],
    'gcn_spmm_csr_add_out_dropout_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _f32, _u64, _u64,
                                         _vp, _i64, _i64, _vp, _sz, _vp],
    'gcn_spmm_csr_dropout_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _f32, _u64, _u64,
                                 _vp, _vp, _sz, _vp],
    'gcn_spmm_csr_out_dropout_f32': [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _f32, _u64, _u64, _vp,
                                     _i64, _i64, _vp, _sz, _vp],
    'gcn_gemm_out_dropout_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i64, _int, _f32, _u64, _u64, _vp,
                                 _i64, _i64, _vp, _vp],
    'gcn_gemm_gated_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i64, _f32, _vp, _i64, _vp],
    'gcn_row_normalize': [_i64, _i64, _vp, _vp, _int, _vp, _vp],
    'gcn_dropout_f32': [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _f32, _u64, _u64, _vp, _u64, _vp],
    'gcn_gemm_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp],
    'gcn_dropout_bits': [_i64, _i64, _f32, _u64, _u64, _vp, _u64, _i64, _vp, _i64, _vp],
    'gcn_gemm_masked_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _int, _vp, _i64, _int, _vp],
    'gcn_gemm_tn_masked_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _i64, _vp, _sz, _vp],
    'gcn_gemm_tn_workspace_bytes': [_i64, _i64, _i64, ctypes.POINTER(ctypes.c_size_t)],
    'gcn_gather_rows_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp],
    'gcn_comm_unique_id': [_vp],
    'gcn_comm_init_rank': [ctypes.POINTER(_vp), _int, _vp, _int],
    'gcn_comm_destroy': [_vp],
    'gcn_allgather_rows_f32': [_vp, _vp, _vp, _i64, _i64, _vp],
    'gcn_exchange_rows_f32': [_vp, _vp, _pi64, _vp, _pi64, _i64, _vp],
    'gcn_softmax_xent_f32': [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _f32, _vp, _vp, _i64, _vp],
    'gcn_loss_summary_workspace_bytes': [_i64, _i64, ctypes.POINTER(ctypes.c_size_t)],
    'gcn_loss_summary_f32': [_i64, _vp, _i64, _vp, _f32, _vp, _sz, _vp, _vp],
    'gcn_adam_f32': [_i64, _vp, _vp, _vp, _vp, _f32, _vp, _f32, _f32, _f32, _f32, _vp],
    'gcn_gemm_tn_f32': [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, ctypes.c_size_t, _vp],
}

_lib = None


class GcnHipError(RuntimeError):
    """A HIP call or kernel launch inside libgcn_hip.so failed."""


def load():
    """Load the shared library (once) and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'gcn_amd: %s is not built. Run `make -C gcn_amd/csrc` (or __graft_entry__.build()). '
            'There is no CPU fallback for the HIP path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name in ('gcn_last_error_string', 'gcn_build_hash') else ctypes.c_int
    _lib = lib
    return lib


def build_hash():
    """sha256 of the sources the loaded library was built from."""
    return load().gcn_build_hash().decode('ascii')


def source_hash():
    """The same hash over the source files on disk: equal to ``build_hash()`` unless the library is stale."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, 'csrc')
    names = sorted(os.path.basename(p) for ext in ('*.hip', '*.cpp', '*.h') for p in glob.glob(os.path.join(csrc, ext)))
    names.remove('asan_check.cpp')                 # the sanitizer driver is not part of the library
    digest = hashlib.sha256()
    for path in [os.path.join(csrc, n) for n in names] + [os.path.join(os.path.dirname(_HERE), 'include', 'gcn_hip.h')]:
        with open(path, 'rb') as fh:
            digest.update(fh.read())
    return digest.hexdigest()


def check(status):
    """Turn a gcn_status into the exception the reference's callers would see."""
    if status == GCN_OK:
        return
    msg = load().gcn_last_error_string().decode('utf-8', 'replace')
    if status in (GCN_ERR_BAD_ARG, GCN_ERR_INDEX_RANGE, GCN_ERR_WORKSPACE):
        raise ValueError('libgcn_hip: ' + msg)
    raise GcnHipError('libgcn_hip: ' + msg)


def call(name, *args):
    check(getattr(load(), name)(*args))
