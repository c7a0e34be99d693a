"""Device-resident CSR matrices and the HIP operators on them.

torch is used for device memory and streams only; every computation is a call
into ``libgcn_hip.so`` through ``gcn_amd._lib``.
"""
import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ValueError('gcn_amd operators take tensors that live on the GPU')


def _workspace(nbytes, device):
    # torch's caching allocator hands out 512-byte aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _rowmajor(x):
    """float32 row-major 2-D view (copy only if the layout forces one)."""
    if x.dim() != 2:
        raise ValueError('expected a 2-D tensor')
    if x.dtype != torch.float32:
        x = x.float()
    if x.shape[1] > 0 and x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x


def _ld(x):
    return x.stride(0) if x.shape[0] > 1 else max(x.shape[1], 1)


GRADIENT_SPLIT = 64                 # split threshold of the plans of small transposed matrices
SMALL_MATRIX_WORK = 8192 * 256      # entries + 4 per row below which 256-entry strips cannot fill the chip


class SpmmPlan(object):
    """Owner of a ``gcn_spmm_plan`` (row-length analysis of one CSR structure)."""

    def __init__(self, n_rows, n_cols, nnz, indptr, split=0):
        """``split``: rows longer than this many entries are summed in chunks (0: the library's
        defaults, under which every row of up to 256 entries keeps the serial order of its additions)."""
        _require_cuda(indptr)
        handle = ctypes.c_void_p()
        _lib.call('gcn_spmm_plan_create_split', n_rows, n_cols, nnz, _p(indptr), int(split), ctypes.byref(handle),
                  _stream())
        self.handle = handle
        info = (ctypes.c_int64 * 4)()
        _lib.call('gcn_spmm_plan_info', self.handle, info)
        self.n_strips, self.n_long_rows, self.n_chunks, self.split_nnz = (int(v) for v in info)
        self._ws = {}

    def set_hot_columns(self, hot_cols):
        """Rows of B from column ``hot_cols`` on are loaded non-temporally by wide products (0: off)."""
        _lib.call('gcn_spmm_plan_set_hot_columns', self.handle, int(hot_cols))

    def set_lds_hub_rows(self, hub_rows):
        """Narrow products keep the first ``hub_rows`` rows of B (as many as fit 32 KiB) in LDS (0: off)."""
        _lib.call('gcn_spmm_plan_set_lds_hub_rows', self.handle, int(hub_rows))

    def set_timing(self, enable=True):
        """Record HIP events around the kernels of every product on this plan."""
        _lib.call('gcn_spmm_plan_set_timing', self.handle, 1 if enable else 0)

    def last_timing(self):
        """(row kernel ms, long-row reduction ms) of the last product; waits for it."""
        ms = (ctypes.c_float * 2)()
        _lib.call('gcn_spmm_plan_last_timing', self.handle, ms)
        return float(ms[0]), float(ms[1])

    def workspace(self, f, device):
        nbytes = ctypes.c_size_t()
        _lib.call('gcn_spmm_plan_workspace_bytes', self.handle, f, ctypes.byref(nbytes))
        if nbytes.value == 0:
            return None, 0
        ws = self._ws.get(nbytes.value)
        if ws is None:
            ws = _workspace(nbytes.value, device)
            self._ws = {nbytes.value: ws}      # keep only the latest size
        return ws, nbytes.value

    def __del__(self):
        try:
            if self.handle:
                _lib.load().gcn_spmm_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceCSR(object):
    """CSR matrix in HBM: ``indptr`` int64 [n_rows+1], ``indices`` int32,
    ``values`` float32.  ``values64`` keeps the float64 values when the matrix
    came out of ``preprocess_adj``; ``perm`` maps every entry to its position in
    the matrix this one was transposed from."""

    def __init__(self, n_rows, n_cols, indptr, indices, values, values64=None, perm=None):
        _require_cuda(indptr, indices, values)
        if indptr.dtype != torch.int64 or indices.dtype != torch.int32 or values.dtype != torch.float32:
            raise ValueError('DeviceCSR wants int64 indptr, int32 indices, float32 values')
        if indptr.numel() != n_rows + 1 or indices.numel() != values.numel():
            raise ValueError('CSR arrays do not match the shape')
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.indptr, self.indices, self.values = indptr.contiguous(), indices.contiguous(), values.contiguous()
        self.values64 = values64
        self.perm = perm
        self._plan = None
        self._t = None

    @property
    def nnz(self):
        return int(self.indices.numel())

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    @property
    def device(self):
        return self.indptr.device

    @property
    def plan(self):
        if self._plan is None:
            self._plan = SpmmPlan(self.n_rows, self.n_cols, self.nnz, self.indptr, getattr(self, 'plan_split', 0))
        return self._plan

    def check_indices(self):
        """Raise ValueError if a column index is out of range (the check the
        reference's TensorFlow kernel makes per nonzero)."""
        _lib.call('gcn_check_indices', self.nnz, _p(self.indices), self.n_cols, _stream())

    def transpose(self, for_gradient=False):
        """Stable transpose (cached): the adjoint operand of the backward products.

        ``for_gradient``: the caller multiplies a gradient with it (X^T . dP, A^T . dZ), a product that is
        compared with the reference to a tolerance.  On a matrix too small to fill the chip the
        transpose's plan then cuts rows at 64 entries -- on a citation graph the rows of X^T are few and
        long (a word's documents): Cora X^T . dP 34 -> 14 us.  The plan is made on first use, so the
        first caller decides."""
        if self._t is None:
            self._t = csr_transpose(self)
        t = self._t
        if for_gradient and t._plan is None and t.nnz + 4 * t.n_rows < SMALL_MATRIX_WORK:
            t.plan_split = GRADIENT_SPLIT
        return t

    @classmethod
    def from_scipy(cls, m, device='cuda'):
        m = m.tocsr()
        return cls(m.shape[0], m.shape[1],
                   torch.as_tensor(m.indptr, dtype=torch.int64).to(device),
                   torch.as_tensor(m.indices, dtype=torch.int32).to(device),
                   torch.as_tensor(m.data, dtype=torch.float32).to(device))


# --------------------------------------------------------------------------- #
# graph preprocessing
# --------------------------------------------------------------------------- #
def coo_to_csr(n_rows, n_cols, rows, cols, vals=None, add_self_loops=False, binary=False):
    """COO (any order) -> (indptr int64, indices int32, values float64 or None).
    Duplicates are summed; ``add_self_loops`` merges in ``sp.eye(n)``."""
    _require_cuda(rows, cols, vals)
    rows = rows.to(torch.int32).contiguous()
    cols = cols.to(torch.int32).contiguous()
    if vals is not None:
        vals = vals.to(torch.float64).contiguous()
    nnz_in = rows.numel()
    dev = rows.device
    flags = (_lib.GCN_ADD_SELF_LOOPS if add_self_loops else 0) | (_lib.GCN_BINARY if binary else 0)
    cap, ws_bytes = ctypes.c_int64(), ctypes.c_size_t()
    _lib.call('gcn_coo_to_csr_bound', n_rows, nnz_in, flags, ctypes.byref(cap), ctypes.byref(ws_bytes))
    indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    indices = torch.empty(cap.value, dtype=torch.int32, device=dev)
    vals_out = None if binary else torch.empty(cap.value, dtype=torch.float64, device=dev)
    nnz_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(ws_bytes.value, dev)
    _lib.call('gcn_coo_to_csr', n_rows, n_cols, nnz_in, _p(rows), _p(cols), _p(vals), flags, _p(indptr), _p(indices),
              _p(vals_out), _p(nnz_out), _p(ws), ws.numel(), _stream())
    nnz = int(nnz_out.item())
    return indptr, indices[:nnz].clone(), (None if binary else vals_out[:nnz].clone())


def rows_to_indptr(n_rows, rows):
    """indptr of a row-grouped triple (rows non-decreasing), order inside rows kept."""
    _require_cuda(rows)
    rows = rows.to(torch.int32).contiguous()
    if rows.numel() > 1 and bool((rows[1:] < rows[:-1]).any()):
        raise ValueError('rows must be non-decreasing')
    indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=rows.device)
    _lib.call('gcn_rows_to_indptr', n_rows, rows.numel(), _p(rows), _p(indptr), _stream())
    return indptr


def row_sums(indptr, vals=None):
    n = indptr.numel() - 1
    out = torch.empty(n, dtype=torch.float64, device=indptr.device)
    _lib.call('gcn_row_sums', n, _p(indptr), _p(vals), _p(out), _stream())
    return out


def degree_scale(rowsum, type='sym'):
    out = torch.empty_like(rowsum)
    _lib.call('gcn_degree_scale', rowsum.numel(), _p(rowsum), _norm_code(type), _p(out), _stream())
    return out


def _norm_code(type):
    if type == 'sym':
        return _lib.GCN_NORM_SYM
    if type == 'rw':
        return _lib.GCN_NORM_RW
    raise ValueError('Invalid normalisation type: ' + str(type))


def normalize_values(indptr, indices, vals, d, type='sym', want64=True):
    """(float64, float32) values of normalize_adj on CSR storage (see gcn_hip.h);
    ``want64=False`` skips the float64 copy."""
    n = indptr.numel() - 1
    nnz = indices.numel()
    out64 = torch.empty(nnz, dtype=torch.float64, device=indptr.device) if want64 else None
    out32 = torch.empty(nnz, dtype=torch.float32, device=indptr.device)
    _lib.call('gcn_normalize_adj', n, nnz, _p(indptr), _p(indices), _p(vals), _p(d), _norm_code(type), _p(out64),
              _p(out32), _stream())
    return out64, out32


def csr_transpose(a):
    """Stable transpose of a DeviceCSR; ``.perm`` of the result gives source positions."""
    dev = a.device
    nnz = a.nnz
    ws_bytes = ctypes.c_size_t()
    _lib.call('gcn_csr_transpose_bound', a.n_rows, a.n_cols, nnz, ctypes.byref(ws_bytes))
    ws = _workspace(ws_bytes.value, dev)
    t_indptr = torch.empty(a.n_cols + 1, dtype=torch.int64, device=dev)
    t_indices = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_vals = torch.empty(nnz, dtype=torch.float32, device=dev)
    perm = torch.empty(nnz, dtype=torch.int64, device=dev)
    _lib.call('gcn_csr_transpose', a.n_rows, a.n_cols, nnz, _p(a.indptr), _p(a.indices), _p(a.values), 4,
              _p(t_indptr), _p(t_indices), _p(t_vals), _p(perm), _p(ws), ws.numel(), _stream())
    if a.perm is not None:
        perm = a.perm[perm]
    return DeviceCSR(a.n_cols, a.n_rows, t_indptr, t_indices, t_vals, perm=perm)


def degree_order(indptr):
    """Vertex order by descending row length (ties by ascending id): ``(new_id, old_id)`` int32 device
    tensors with ``new_id[v]`` the position of vertex v and ``old_id[i]`` the vertex at position i."""
    _require_cuda(indptr)
    n = indptr.numel() - 1
    ws_bytes = ctypes.c_size_t()
    _lib.call('gcn_degree_order_bound', n, ctypes.byref(ws_bytes))
    ws = _workspace(ws_bytes.value, indptr.device)
    new_id = torch.empty(n, dtype=torch.int32, device=indptr.device)
    old_id = torch.empty(n, dtype=torch.int32, device=indptr.device)
    _lib.call('gcn_degree_order', n, _p(indptr), _p(new_id), _p(old_id), _p(ws), ws.numel(), _stream())
    return new_id, old_id


def hub_order(indptr, n_hubs):
    """Hub-first vertex order: the ``n_hubs`` vertices of highest degree first (in degree order), every other
    vertex after them in its old order.  ``(new_id, old_id)`` as for ``degree_order``."""
    _require_cuda(indptr)
    n = indptr.numel() - 1
    ws_bytes = ctypes.c_size_t()
    _lib.call('gcn_hub_order_bound', n, ctypes.byref(ws_bytes))
    ws = _workspace(ws_bytes.value, indptr.device)
    new_id = torch.empty(n, dtype=torch.int32, device=indptr.device)
    old_id = torch.empty(n, dtype=torch.int32, device=indptr.device)
    _lib.call('gcn_hub_order', n, _p(indptr), int(n_hubs), _p(new_id), _p(old_id), _p(ws), ws.numel(), _stream())
    return new_id, old_id


def permute_rows(a, old_id):
    """The CSR matrix whose row i is row ``old_id[i]`` of ``a``, the entries of every row in their old order
    (the accumulation order of a product does not change); ``.perm`` gives every entry's position in the
    matrix ``a`` came from, so that a fused sparse dropout draws the decisions of the unpermuted matrix."""
    old = old_id.to(torch.int64)
    lens = (a.indptr[1:] - a.indptr[:-1])[old]
    indptr = torch.zeros(a.n_rows + 1, dtype=torch.int64, device=a.device)
    torch.cumsum(lens, 0, out=indptr[1:])
    src = torch.repeat_interleave(a.indptr[:-1][old] - indptr[:-1], lens) + torch.arange(a.nnz, device=a.device)
    perm = src if a.perm is None else a.perm[src]
    return DeviceCSR(a.n_rows, a.n_cols, indptr, a.indices[src].contiguous(), a.values[src].contiguous(), perm=perm)


def csr_relabel(n_rows, n_cols, indptr, indices, vals=None, new_row=None, new_col=None):
    """(indptr, indices, vals) of the matrix with entry (r, c) moved to (new_row[r], new_col[c]);
    ``vals`` float64 or None (structure only)."""
    _require_cuda(indptr, indices, vals, new_row, new_col)
    dev = indptr.device
    nnz = indices.numel()
    if vals is not None:
        vals = vals.to(torch.float64).contiguous()
    for lab in (new_row, new_col):
        if lab is not None and (lab.dtype != torch.int32 or not lab.is_contiguous()):
            raise ValueError('labels must be contiguous int32 tensors')
    if (new_row is not None and new_row.numel() != n_rows) or (new_col is not None and new_col.numel() != n_cols):
        raise ValueError('one label per row / column')
    ws_bytes = ctypes.c_size_t()
    _lib.call('gcn_csr_relabel_bound', n_rows, nnz, 0 if vals is None else 1, ctypes.byref(ws_bytes))
    ws = _workspace(ws_bytes.value, dev)
    indptr_out = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    indices_out = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    vals_out = None if vals is None else torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    nnz_out = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.call('gcn_csr_relabel', n_rows, n_cols, nnz, _p(indptr), _p(indices.contiguous()), _p(vals), _p(new_row),
              _p(new_col), _p(indptr_out), _p(indices_out), _p(vals_out), _p(nnz_out), _p(ws), ws.numel(), _stream())
    m = int(nnz_out.item())
    del ws
    return indptr_out, indices_out[:m], (None if vals is None else vals_out[:m])


def triple_to_device_csr(triple, device='cuda'):
    """Device CSR of a fed ``(indices [nnz,2], values, dense_shape)`` triple, values cast to
    float32 as the float32 sparse placeholder does (train.py:145-146).  The order of the entries
    inside every row is kept (it is the accumulation order of the reference's kernel); a triple
    that is not row-grouped is stably regrouped by row first."""
    import numpy as np
    coords = np.asarray(triple[0])
    values = np.asarray(triple[1])
    n_rows, n_cols = (int(v) for v in triple[2])
    if coords.ndim != 2 or coords.shape[1] != 2 or coords.shape[0] != values.shape[0]:
        raise ValueError('a sparse triple needs indices [nnz, 2] and values [nnz]')
    rows = coords[:, 0]
    if rows.size and np.any(rows[1:] < rows[:-1]):
        order = np.argsort(rows, kind='stable')
        coords, values, rows = coords[order], values[order], rows[order]
    if rows.size and (rows[0] < 0 or rows[-1] >= n_rows):
        raise ValueError('a row index is outside [0, %d)' % n_rows)
    d_rows = torch.as_tensor(np.ascontiguousarray(rows).astype(np.int32)).to(device)
    indptr = rows_to_indptr(n_rows, d_rows)
    csr = DeviceCSR(n_rows, n_cols, indptr,
                    torch.as_tensor(np.ascontiguousarray(coords[:, 1]).astype(np.int32)).to(device),
                    torch.as_tensor(values.astype(np.float32)).to(device))
    csr.check_indices()
    return csr


# --------------------------------------------------------------------------- #
# synthetic inputs
# --------------------------------------------------------------------------- #
def rmat_csr(scale, edge_factor=16, seed=42, self_loops=False, device='cuda', scramble=True):
    """CSR structure (indptr int64, indices int32) of the symmetrised, deduplicated R-MAT graph;
    with ``self_loops`` the structure of A + I.  ``scramble=False`` keeps the generator's own vertex ids."""
    flags = (_lib.GCN_ADD_SELF_LOOPS if self_loops else 0) | (0 if scramble else _lib.GCN_RMAT_NATURAL_ORDER)
    cap, ws_bytes = ctypes.c_int64(), ctypes.c_size_t()
    _lib.call('gcn_rmat_csr_bound', scale, edge_factor, flags, ctypes.byref(cap), ctypes.byref(ws_bytes))
    n = 1 << scale
    indptr = torch.empty(n + 1, dtype=torch.int64, device=device)
    indices = torch.empty(cap.value, dtype=torch.int32, device=device)
    nnz_out = torch.zeros(1, dtype=torch.int64, device=device)
    ws = _workspace(ws_bytes.value, indptr.device)
    _lib.call('gcn_rmat_csr', scale, edge_factor, seed, flags, _p(indptr), _p(indices), _p(nnz_out), _p(ws),
              ws.numel(), _stream())
    nnz = int(nnz_out.item())
    del ws
    return indptr, indices[:nnz].clone()


def rmat_csr_rows(scale, row_begin, row_end, edge_factor=16, seed=42, self_loops=False, device='cuda',
                  scramble=True):
    """The rows [row_begin, row_end) of ``rmat_csr``'s graph (indptr from 0, global column indices)
    without building the other rows: the shard of one rank."""
    flags = (_lib.GCN_ADD_SELF_LOOPS if self_loops else 0) | (0 if scramble else _lib.GCN_RMAT_NATURAL_ORDER)
    n_keys, ws_bytes = ctypes.c_int64(), ctypes.c_size_t()
    with torch.cuda.device(device):
        _lib.call('gcn_rmat_rows_count', scale, edge_factor, seed, flags, row_begin, row_end, ctypes.byref(n_keys),
                  _stream())
        _lib.call('gcn_rmat_csr_rows_bound', n_keys.value, ctypes.byref(ws_bytes))
        indptr = torch.empty(row_end - row_begin + 1, dtype=torch.int64, device=device)
        indices = torch.empty(max(n_keys.value, 1), dtype=torch.int32, device=device)
        nnz_out = torch.zeros(1, dtype=torch.int64, device=device)
        ws = _workspace(ws_bytes.value, indptr.device)
        _lib.call('gcn_rmat_csr_rows', scale, edge_factor, seed, flags, row_begin, row_end, n_keys.value, _p(indptr),
                  _p(indices), _p(nnz_out), _p(ws), ws.numel(), _stream())
        nnz = int(nnz_out.item())
    del ws
    return indptr, indices[:nnz].clone()


def synthetic_dense(n, f, seed=7, device='cuda', first_row=0):
    """Rows first_row .. first_row + n - 1 of the seeded synthetic matrix with f columns."""
    out = torch.empty((n, f), dtype=torch.float32, device=device)
    _lib.call('gcn_fill_synthetic_range_f32', first_row * f, n * f, seed, _p(out), _stream())
    return out


# --------------------------------------------------------------------------- #
# products
# --------------------------------------------------------------------------- #
def _check_out(out, n_rows, f):
    """An output may be a column slice of a wider row-major matrix (its row stride is passed as ldc)."""
    _require_cuda(out)
    if tuple(out.shape) != (n_rows, f) or out.dtype != torch.float32 or (f > 1 and out.stride(1) != 1) \
            or (n_rows > 1 and out.stride(0) < f):
        raise ValueError('out must be a row-major float32 [%d x %d] view' % (n_rows, f))


class OutDropout(object):
    """Dropout of a product's OUTPUT, applied in its store (the next layer's tf.nn.dropout on its input,
    gcn/layers.py:185).  Element (r, c) of the output draws decision ``index_offset + r * pitch + c`` of
    stream (seed, stream_id): the one ``dropout`` draws for a matrix ``pitch`` columns wide of which
    the output is a row block (index_offset = first row * pitch) and possibly a column range."""

    def __init__(self, keep_prob, seed, stream_id, index_offset=0, pitch=None, row_ids=None):
        """``row_ids`` (int64 device tensor, dense products only): the output holds a subset of the rows of
        that matrix, and its row r draws the decisions of row ``row_ids[r]``."""
        self.keep_prob, self.seed, self.stream_id = float(keep_prob), int(seed), int(stream_id)
        self.index_offset, self.pitch, self.row_ids = int(index_offset), pitch, row_ids

    def columns(self, c0):
        """The same decisions for the column range that starts at column ``c0``."""
        return OutDropout(self.keep_prob, self.seed, self.stream_id, self.index_offset + c0, self.pitch, self.row_ids)


def _two_part_form(a, f, first, b, out):
    """True when the product kernels can read the operand [first; b] in place: 128-bit rows in both parts and
    a matrix large enough for the strip instances that have the two-part form."""
    if f % 4 or a.nnz + 4 * a.n_rows < SMALL_MATRIX_WORK:
        return False
    for t in (first, b, out):
        if t is not None and t.numel() and (t.data_ptr() % 16 or _ld(t) % 4):
            return False
    return True


def spmm(a, b, relu=False, out=None, wide_split=False, out_dropout=None, first=None):
    """``a @ b`` for a DeviceCSR ``a`` and dense float32 ``b`` (optionally with ReLU fused).
    ``wide_split`` cuts long rows as a wide product does, whatever the width of ``b``: use it when
    ``b`` is a range of the columns of a wider operand and the bits must match the full product.
    ``out_dropout`` (an OutDropout): the result is written already dropped, after the ReLU.
    ``first``: the operand is ``[first; b]`` -- the rows of ``first`` followed by those of ``b`` (which may be
    None or empty) -- and is read in place, without stacking the two (a stacked copy is made only for
    layouts the kernels' two-part form does not take)."""
    if first is not None:
        _require_cuda(first)
        first = _rowmajor(first)
        if b is None:
            b = first.new_empty((0, first.shape[1]))
        _require_cuda(b)
        b = _rowmajor(b)
        if first.shape[0] + b.shape[0] != a.n_cols or first.shape[1] != b.shape[1]:
            raise ValueError('shape mismatch: %s @ [%s; %s]' % (a.shape, tuple(first.shape), tuple(b.shape)))
        if first.shape[0] == 0:
            first = None
        elif not _two_part_form(a, b.shape[1], first, b, out):
            b, first = torch.cat([first, b]), None
    _require_cuda(b)
    b = _rowmajor(b)
    if first is None and b.shape[0] != a.n_cols:
        raise ValueError('shape mismatch: %s @ %s' % (a.shape, tuple(b.shape)))
    f = b.shape[1]
    if out is None:
        out = torch.empty((a.n_rows, f), dtype=torch.float32, device=b.device)
    else:
        _check_out(out, a.n_rows, f)
    ws, ws_bytes = a.plan.workspace(f, b.device)
    flags = (_lib.GCN_RELU if relu else 0) | (_lib.GCN_SPLIT_WIDE if wide_split else 0)
    if first is not None:
        d = out_dropout if (out_dropout is not None and out_dropout.keep_prob < 1.0) else None
        if d is not None and d.row_ids is not None:
            raise ValueError('row_ids apply to dense products only')
        rest = b if b.shape[0] else None
        _lib.call('gcn_spmm_csr_two_part_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(first),
                  _ld(first), first.shape[0], _p(rest), _ld(b) if rest is not None else f, _p(out), _ld(out), flags,
                  d.keep_prob if d else 1.0, d.seed if d else 0, d.stream_id if d else 0,
                  _p(_stream_offset) if d else ctypes.c_void_p(0), d.index_offset if d else 0,
                  (f if d.pitch is None else int(d.pitch)) if d else f, _p(ws), ws_bytes, _stream())
        return out
    if out_dropout is not None and out_dropout.keep_prob < 1.0:
        d = out_dropout
        if d.row_ids is not None:
            raise ValueError('row_ids apply to dense products only')
        _lib.call('gcn_spmm_csr_out_dropout_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(b),
                  _ld(b), _p(out), _ld(out), flags, d.keep_prob, d.seed, d.stream_id, _p(_stream_offset),
                  d.index_offset, f if d.pitch is None else int(d.pitch), _p(ws), ws_bytes, _stream())
        return out
    _lib.call('gcn_spmm_csr_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(b), _ld(b),
              _p(out), _ld(out), flags, _p(ws), ws_bytes, _stream())
    return out


NOALLOC_SCRATCH_BYTES = 32 << 20


def spmm_noalloc(a, b, out, relu=False):
    """``a @ b`` into ``out`` through the entry point that allocates nothing: the partial rows of long
    rows go to the library's own 32 MiB array instead of a workspace (ValueError when the product
    needs more).  ``b`` and ``out`` must already be row-major float32 device tensors; the plan of ``a``
    must exist (touch ``a.plan`` beforehand: creating it allocates)."""
    _require_cuda(b, out)
    if b.dim() != 2 or b.dtype != torch.float32 or (b.shape[1] > 1 and b.stride(1) != 1):
        raise ValueError('spmm_noalloc takes a row-major float32 operand as it is')
    if b.shape[0] != a.n_cols:
        raise ValueError('shape mismatch: %s @ %s' % (a.shape, tuple(b.shape)))
    if a._plan is None:
        raise ValueError('create the plan first (a.plan): plan creation allocates')
    f = b.shape[1]
    _check_out(out, a.n_rows, f)
    _lib.call('gcn_spmm_csr_f32_noalloc', a._plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(b), _ld(b),
              _p(out), _ld(out), _lib.GCN_RELU if relu else 0, _stream())
    return out


def spmm_add(a, b, d, relu=False, out=None, wide_split=False, out_dropout=None):
    """``a @ b + d`` with the addition fused into the product's epilogue (one smoothing step): every row is
    summed as ``spmm`` sums it and ``d``'s row is added last.  ``out`` may be ``d`` itself (each element is read
    and then written by the same lane).  ``wide_split`` and ``out_dropout`` as in ``spmm``."""
    _require_cuda(b, d)
    b, d = _rowmajor(b), _rowmajor(d)
    if b.shape[0] != a.n_cols or d.shape != (a.n_rows, b.shape[1]):
        raise ValueError('shape mismatch in spmm_add')
    f = b.shape[1]
    if out is None:
        out = torch.empty((a.n_rows, f), dtype=torch.float32, device=b.device)
    else:
        _check_out(out, a.n_rows, f)
    ws, ws_bytes = a.plan.workspace(f, b.device)
    flags = (_lib.GCN_RELU if relu else 0) | (_lib.GCN_SPLIT_WIDE if wide_split else 0)
    if out_dropout is not None and out_dropout.keep_prob < 1.0:
        o = out_dropout
        if o.row_ids is not None:
            raise ValueError('row_ids apply to dense products only')
        _lib.call('gcn_spmm_csr_add_out_dropout_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f,
                  _p(b), _ld(b), _p(d), _ld(d), _p(out), _ld(out), flags, o.keep_prob, o.seed, o.stream_id,
                  _p(_stream_offset), o.index_offset, f if o.pitch is None else int(o.pitch), _p(ws), ws_bytes, _stream())
        return out
    _lib.call('gcn_spmm_csr_add_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(b), _ld(b),
              _p(d), _ld(d), _p(out), _ld(out), flags, _p(ws), ws_bytes, _stream())
    return out


class PackedRows(object):
    """A dense [n_rows x f] matrix (f <= 256) without its zeros: per row eight {mask word, kept columns before it}
    pairs and the kept values in column order, at a pitch of ``ld`` floats (``data``: [n_rows x ld] float32).  What
    ``spmm_packed`` gathers instead of dense rows: 64 + 4 k bytes for a row with k kept values."""

    MAX_F = 256

    def __init__(self, n_rows, f, data):
        self.n_rows, self.f, self.data = int(n_rows), int(f), data

    @property
    def ld(self):
        return self.data.shape[1]

    @staticmethod
    def pitch(f):
        out = ctypes.c_int64()
        _lib.call('gcn_pack_pitch', int(f), ctypes.byref(out))
        return int(out.value)

    @classmethod
    def empty(cls, n_rows, f, device):
        # torch's allocator returns 512-byte aligned blocks; rows are a multiple of 128 bytes
        return cls(n_rows, f, torch.empty((n_rows, cls.pitch(f)), dtype=torch.float32, device=device))

    def rows(self, begin, end):
        """Rows begin .. end - 1 as PackedRows over the same memory."""
        return PackedRows(end - begin, self.f, self.data[begin:end])

    def dense(self, out=None):
        """The expanded matrix."""
        if out is None:
            out = torch.empty((self.n_rows, self.f), dtype=torch.float32, device=self.data.device)
        _lib.call('gcn_unpack_rows_f32', self.n_rows, self.f, _p(self.data), self.ld, _p(out), _ld(out), _stream())
        return out

    def kept(self):
        """Number of stored values per row (int64)."""
        words = self.data[:, :16:2].contiguous().view(torch.int32)
        total = torch.zeros(self.n_rows, dtype=torch.int64, device=self.data.device)
        for b in range(32):
            total += ((words >> b) & 1).sum(dim=1)
        return total


def pack_rows(x, out=None):
    """The nonzeros of ``x`` [n x f], f <= 256, as PackedRows."""
    _require_cuda(x)
    x = _rowmajor(x)
    n, f = x.shape
    out = out or PackedRows.empty(n, f, x.device)
    _lib.call('gcn_pack_rows_f32', n, f, _p(x), _ld(x), _p(out.data), out.ld, _stream())
    return out


def dropout_pack(x, keep_prob, seed, stream_id, index_offset=0, out=None, row_ids=None):
    """``dropout(x, keep_prob, seed, stream_id, index_offset=...)`` as PackedRows, the dense result never written.
    ``row_ids`` (int64, one per row of ``x``): row r of ``x`` draws the decisions of row ``row_ids[r]`` of the matrix
    the stream belongs to (rows of another rank's block held here)."""
    _require_cuda(x)
    x = _rowmajor(x)
    n, f = x.shape
    out = out or PackedRows.empty(n, f, x.device)
    if out.n_rows != n or out.f != f:
        raise ValueError('the packed output holds %d x %d, the input is %d x %d' % (out.n_rows, out.f, n, f))
    if row_ids is not None:
        _require_cuda(row_ids)
        if row_ids.dtype != torch.int64 or row_ids.numel() != n or not row_ids.is_contiguous():
            raise ValueError('row_ids must be a contiguous int64 tensor with one entry per row')
    _lib.call('gcn_dropout_pack_rows_f32', n, f, _p(x), _ld(x), float(keep_prob), int(seed), int(stream_id),
              _p(_stream_offset), int(index_offset), f, _p(row_ids) if row_ids is not None else None,
              _p(out.data), out.ld, _stream())
    return out


GATED_PACK_MAX_K = 16


def gemm_gated_pack(a, b, gate, divisor, out=None, rows=None):
    """``where(gate > 0, (a @ b.T) / divisor, 0)`` as PackedRows, for ``a`` [m x k], ``b`` [n x k] with n <= 256 and
    k <= 16: ``gemm_gated(a, b, gate, divisor, trans_b=True)`` without the dense result (equal to it up to the
    order of the k additions).  ``rows`` (ascending int64 row numbers): ``a`` is zero in every other row -- those
    rows come out empty without being read, only the listed rows are formed."""
    _require_cuda(a, b, gate, rows)
    a, b, gate = _rowmajor(a), _rowmajor(b), _rowmajor(gate)
    m, k = a.shape
    n = b.shape[0]
    if b.shape[1] != k or tuple(gate.shape) != (m, n):
        raise ValueError('shape mismatch in gemm_gated_pack')
    out = out or PackedRows.empty(m, n, a.device)
    if rows is not None and (rows.dtype != torch.int64 or not rows.is_contiguous() or rows.dim() != 1):
        raise ValueError('rows must be a contiguous 1-d int64 tensor')
    listed = 0 if rows is None else rows.numel()
    if rows is not None and listed == 0:         # an empty list is still a list (an empty tensor has no address)
        rows = torch.zeros(1, dtype=torch.int64, device=a.device)
    _lib.call('gcn_gemm_gated_pack_rows_f32', m, n, k, _p(a), _ld(a), _p(b), _ld(b), _p(gate), _ld(gate),
              float(divisor), _p(rows), listed, _p(out.data), out.ld, _stream())
    return out


def spmm_packed(a, b, relu=False, out=None):
    """``a @ b.dense()`` for a PackedRows ``b``: the values of ``spmm(a, b.dense())`` (long rows cut as the wide
    product cuts them), gathering packed rows."""
    if b.n_rows != a.n_cols:
        raise ValueError('shape mismatch: %s @ packed [%d x %d]' % (a.shape, b.n_rows, b.f))
    f = b.f
    if out is None:
        out = torch.empty((a.n_rows, f), dtype=torch.float32, device=b.data.device)
    else:
        _check_out(out, a.n_rows, f)
    ws, ws_bytes = a.plan.workspace(f, b.data.device)
    _lib.call('gcn_spmm_csr_packed_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), f, _p(b.data), b.ld,
              _p(out), _ld(out), _lib.GCN_RELU if relu else 0, _p(ws), ws_bytes, _stream())
    return out


def row_normalize(indptr, vals):
    """Bag-of-words scaling r_inv[row] * value (gcn/utils.py:279-285) of CSR values (float32 or float64)."""
    _require_cuda(indptr, vals)
    if vals.dtype not in (torch.float32, torch.float64):
        raise ValueError('values must be float32 or float64')
    vals = vals.contiguous()
    out = torch.empty_like(vals)
    _lib.call('gcn_row_normalize', indptr.numel() - 1, vals.numel(), _p(indptr), _p(vals), vals.element_size(),
              _p(out), _stream())
    return out


# When set (a 1-element int64 device tensor), the dropout kernels add its value to the stream id at
# run time: launches recorded in a HIP graph then draw new masks on every replay.
_stream_offset = None


class stream_offset(object):
    """Context manager: dropout launches inside it take their stream id offset from ``tensor``."""

    def __init__(self, tensor):
        if tensor is not None and (not tensor.is_cuda or tensor.dtype != torch.int64 or tensor.numel() != 1):
            raise ValueError('stream offset must be a 1-element int64 device tensor')
        self.tensor = tensor

    def __enter__(self):
        global _stream_offset
        self.previous, _stream_offset = _stream_offset, self.tensor
        return self.tensor

    def __exit__(self, *exc):
        global _stream_offset
        _stream_offset = self.previous
        return False


def spmm_dropout(a, b, keep_prob, seed, stream_id, relu=False, out=None):
    """``sparse_dropout(a, keep_prob) @ b`` with the dropout fused into the product.
    Entry ids for the random stream come from ``a.perm`` when ``a`` is a transposed copy."""
    _require_cuda(b)
    b = _rowmajor(b)
    if b.shape[0] != a.n_cols:
        raise ValueError('shape mismatch: %s @ %s' % (a.shape, tuple(b.shape)))
    f = b.shape[1]
    if out is None:
        out = torch.empty((a.n_rows, f), dtype=torch.float32, device=b.device)
    ws, ws_bytes = a.plan.workspace(f, b.device)
    _lib.call('gcn_spmm_csr_dropout_f32', a.plan.handle, _p(a.indptr), _p(a.indices), _p(a.values), _p(a.perm), f,
              _p(b), _ld(b), _p(out), _ld(out), _lib.GCN_RELU if relu else 0, float(keep_prob), int(seed),
              int(stream_id), _p(_stream_offset), _p(ws), ws_bytes, _stream())
    return out


def dropout(x, keep_prob, seed, stream_id, gate=None, index_offset=0):
    """tf.nn.dropout(x, keep_prob); with ``gate`` the result is also zeroed where gate <= 0.
    ``index_offset`` shifts the element ids of the random stream (row_begin * width for a row shard)."""
    _require_cuda(x, gate)
    x = _rowmajor(x)
    if gate is not None:
        gate = _rowmajor(gate)
    y = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.call('gcn_dropout_f32', x.shape[0], x.shape[1], _p(x), _ld(x), _p(gate), 0 if gate is None else _ld(gate),
              _p(y), _ld(y), float(keep_prob), int(seed), int(stream_id), _p(_stream_offset), int(index_offset),
              _stream())
    return y


def dropout_bits(n_rows, f, keep_prob, seed, stream_id, index_offset=0, pitch=None, device='cuda'):
    """The decisions ``dropout`` would draw for a matrix [n_rows x f], one bit per element: an int32 tensor
    [n_rows x ceil(f / 32)] whose bit c % 32 of word c / 32 of row r is 1 iff element (r, c) is kept.
    ``gemm_masked`` / ``gemm_tn_masked`` apply them to their left operand as they load it."""
    words = (int(f) + 31) // 32
    bits = torch.empty((int(n_rows), words), dtype=torch.int32, device=device)
    _lib.call('gcn_dropout_bits', int(n_rows), int(f), float(keep_prob), int(seed), int(stream_id), _p(_stream_offset),
              int(index_offset), int(f if pitch is None else pitch), _p(bits), words, _stream())
    return bits


def _check_bits(bits, m, k):
    _require_cuda(bits)
    if bits.dtype != torch.int32 or bits.dim() != 2 or bits.shape[0] != m or bits.shape[1] * 32 < k \
            or (bits.shape[1] > 1 and bits.stride(1) != 1):
        raise ValueError('mask bits must be an int32 [%d x >= %d] tensor' % (m, (k + 31) // 32))
    return bits.stride(0) if m > 1 else bits.shape[1]


def gemm_masked(a, bits, keep_prob, b, trans_b=False, relu=False, out=None):
    """``dropout(a) @ b`` with the dropout applied to ``a`` as the product loads it (``bits`` from
    ``dropout_bits``): the bits of ``gemm(dropout(a, ...), b)`` without the pass that writes the dropped matrix."""
    _require_cuda(a, b)
    a, b = _rowmajor(a), _rowmajor(b)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != k:
        raise ValueError('shape mismatch in gemm_masked')
    ld_words = _check_bits(bits, m, k)
    if out is None:
        c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    else:
        _check_out(out, m, n)
        c = out
    _lib.call('gcn_gemm_masked_f32', m, n, k, _p(a), _ld(a), _p(bits), ld_words, float(keep_prob), _p(b), _ld(b),
              1 if trans_b else 0, _p(c), _ld(c), _lib.GCN_RELU if relu else 0, _stream())
    return c


def gemm_tn_masked(a, bits, keep_prob, g, out=None):
    """``dropout(a).T @ g`` with the same mask bits: the weight gradient of the layer ``gemm_masked`` computes."""
    _require_cuda(a, g)
    a, g = _rowmajor(a), _rowmajor(g)
    m, k = a.shape
    if g.shape[0] != m:
        raise ValueError('shape mismatch in gemm_tn_masked')
    n = g.shape[1]
    ld_words = _check_bits(bits, m, k)
    if out is None:
        out = torch.empty((k, n), dtype=torch.float32, device=a.device)
    else:
        _check_out(out, k, n)
    nbytes = ctypes.c_size_t()
    _lib.call('gcn_gemm_tn_workspace_bytes', m, n, k, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, a.device) if nbytes.value else None
    _lib.call('gcn_gemm_tn_masked_f32', m, n, k, _p(a), _ld(a), _p(bits), ld_words, float(keep_prob), _p(g), _ld(g),
              _p(out), _ld(out), _p(ws), nbytes.value, _stream())
    return out


def gemm(a, b, trans_b=False, relu=False, out=None, out_dropout=None):
    """``a @ b`` (or ``a @ b.T``) in float32 on the MFMA pipe.  ``out_dropout`` (an OutDropout): the result
    is written already dropped, after the ReLU."""
    _require_cuda(a, b)
    a, b = _rowmajor(a), _rowmajor(b)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != k:
        raise ValueError('shape mismatch in gemm')
    if out is None:
        c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    else:
        _check_out(out, m, n)
        c = out
    if out_dropout is not None and out_dropout.keep_prob < 1.0:
        d = out_dropout
        if d.row_ids is not None and (d.row_ids.dtype != torch.int64 or d.row_ids.numel() != m or not d.row_ids.is_cuda):
            raise ValueError('row_ids must be an int64 device tensor with one entry per output row')
        _lib.call('gcn_gemm_out_dropout_f32', m, n, k, _p(a), _ld(a), _p(b), _ld(b), 1 if trans_b else 0, _p(c), _ld(c),
                  _lib.GCN_RELU if relu else 0, d.keep_prob, d.seed, d.stream_id, _p(_stream_offset), d.index_offset,
                  n if d.pitch is None else int(d.pitch), _p(d.row_ids), _stream())
        return c
    _lib.call('gcn_gemm_f32', m, n, k, _p(a), _ld(a), _p(b), _ld(b), 1 if trans_b else 0, _p(c), _ld(c),
              _lib.GCN_RELU if relu else 0, _stream())
    return c


def gemm_gated(a, b, gate, divisor, trans_b=False, out=None):
    """``where(gate > 0, (a @ b) / divisor, 0)``: the gradient of a product's left operand with the
    backward of the dropout(relu(.)) that made that operand folded into the store (``gate`` is the
    dropped activation matrix itself, ``divisor`` the keep probability)."""
    _require_cuda(a, b, gate)
    a, b, gate = _rowmajor(a), _rowmajor(b), _rowmajor(gate)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != k or tuple(gate.shape) != (m, n):
        raise ValueError('shape mismatch in gemm_gated')
    if out is None:
        c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    else:
        _check_out(out, m, n)
        c = out
    _lib.call('gcn_gemm_gated_f32', m, n, k, _p(a), _ld(a), _p(b), _ld(b), 1 if trans_b else 0, _p(gate), _ld(gate),
              float(divisor), _p(c), _ld(c), _stream())
    return c


def gemm_tn(a, g, out=None):
    """``a.T @ g`` for row-major ``a`` [m x k] and ``g`` [m x n] without forming the transpose: the
    weight gradient of a layer, a sum over all m nodes."""
    _require_cuda(a, g)
    a, g = _rowmajor(a), _rowmajor(g)
    m, k = a.shape
    if g.shape[0] != m:
        raise ValueError('shape mismatch in gemm_tn')
    n = g.shape[1]
    if out is None:
        out = torch.empty((k, n), dtype=torch.float32, device=a.device)
    else:
        _check_out(out, k, n)
    nbytes = ctypes.c_size_t()
    _lib.call('gcn_gemm_tn_workspace_bytes', m, n, k, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, a.device) if nbytes.value else None
    _lib.call('gcn_gemm_tn_f32', m, n, k, _p(a), _ld(a), _p(g), _ld(g), _p(out), _ld(out), _p(ws), nbytes.value,
              _stream())
    return out


def softmax_xent(logits, labels, mask, mask_mean, want_grad=True):
    """Per-row (masked cross-entropy, masked correctness) [n x 2] and, with ``want_grad``, the gradient
    of the mean loss w.r.t. the logits.  ``labels`` int32 one-hot [n x c], ``mask`` int32 [n]."""
    _require_cuda(logits, labels, mask)
    logits = _rowmajor(logits)
    if labels.dtype != torch.int32 or mask.dtype != torch.int32:
        raise ValueError('labels and mask must be int32')
    labels = labels.contiguous()
    mask = mask.contiguous()
    n, c = logits.shape
    if tuple(labels.shape) != (n, c) or mask.numel() != n:
        raise ValueError('labels / mask do not match the logits')
    stats = torch.empty((n, 2), dtype=torch.float32, device=logits.device)
    grad = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_grad else None
    _lib.call('gcn_softmax_xent_f32', n, c, _p(logits), _ld(logits), _p(labels), c, _p(mask), float(mask_mean),
              _p(stats), _p(grad), c, _stream())
    return stats, grad


def loss_summary(stats, weights_flat, weight_decay):
    """[loss, accuracy] of a step as a 2-element device tensor: the means of the two columns of
    ``stats`` (from softmax_xent) plus weight_decay * sum(w^2) / 2 over ``weights_flat`` (all variables
    in one buffer; None or empty: no weight-decay term)."""
    _require_cuda(stats)
    n = stats.shape[0]
    if stats.dtype != torch.float32 or tuple(stats.shape) != (n, 2) or not stats.is_contiguous():
        raise ValueError('stats must be a contiguous float32 [n x 2] tensor')
    n_w = 0 if weights_flat is None else weights_flat.numel()
    if n_w and (weights_flat.dtype != torch.float32 or not weights_flat.is_contiguous()):
        raise ValueError('weights must be one contiguous float32 buffer')
    need = ctypes.c_size_t()
    _lib.call('gcn_loss_summary_workspace_bytes', n, n_w, ctypes.byref(need))
    ws = _workspace(need.value, stats.device) if need.value else None
    out = torch.empty(2, dtype=torch.float32, device=stats.device)
    _lib.call('gcn_loss_summary_f32', n, _p(stats), n_w, _p(weights_flat) if n_w else None, float(weight_decay),
              _p(ws), need.value, _p(out), _stream())
    return out


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0):
    """In-place Adam update of one contiguous float32 variable; ``lr_t`` is a float or a 0-d device tensor."""
    _require_cuda(param, grad, m, v)
    if not (param.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or param.dtype != torch.float32:
        raise ValueError('adam_step wants contiguous float32 state')
    grad = grad.contiguous()
    if grad.numel() != param.numel():
        raise ValueError('gradient does not match the variable')
    on_device = isinstance(lr_t, torch.Tensor)
    _lib.call('gcn_adam_f32', param.numel(), _p(param), _p(grad), _p(m), _p(v), 0.0 if on_device else float(lr_t),
              _p(lr_t) if on_device else ctypes.c_void_p(0), float(beta1), float(beta2), float(epsilon),
              float(weight_decay), _stream())
    return param


def gather_rows(src, rows, out):
    """``out[i] = src[rows[i]]`` for a row-major float32 matrix (or a column slice of one) and int64 row ids."""
    _require_cuda(src, rows, out)
    if src.dim() != 2 or src.dtype != torch.float32 or (src.shape[1] > 1 and src.stride(1) != 1):
        raise ValueError('gather_rows wants a row-major float32 matrix')
    if rows.dtype != torch.int64 or not rows.is_contiguous():
        raise ValueError('row ids must be contiguous int64')
    _check_out(out, rows.numel(), src.shape[1])
    _lib.call('gcn_gather_rows_f32', rows.numel(), src.shape[1], src.shape[0], _p(src), _ld(src), _p(rows), _p(out),
              _ld(out), _stream())
    return out
