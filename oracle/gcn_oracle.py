"""CPU restatement of the reference's GCN propagation path (numpy / scipy).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker.

Parity status: the graph half (``normalize_adj``, ``preprocess_adj``,
``sparse_to_tuple``, ``preprocess_features``, ``load_data``) is PINNED -- it is
checked against vectors produced by running the reference's own
``gcn/utils.py`` (see ``make_fixtures.py``).  The TensorFlow half (dropout,
``dot``, layer, model, loss, Adam) is PARITY UNPINNED: TensorFlow cannot be
imported here and the reference holds no tests or golden vectors, so those
functions restate the arithmetic the cited lines ask TensorFlow to do.

Each function cites the reference lines it follows (paths relative to the
reference root).
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------- #
# counter-based random numbers shared bit-for-bit with the HIP kernels
# --------------------------------------------------------------------------- #
_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)
_M1 = _U64(0xBF58476D1CE4E5B9)
_M2 = _U64(0x94D049BB133111EB)


def sm64(x):
    """splitmix64 output function applied to state ``x`` (array or scalar), uint64."""
    with np.errstate(over='ignore'):
        z = np.asarray(x, dtype=np.uint64) + _GOLD
        z = (z ^ (z >> _U64(30))) * _M1
        z = (z ^ (z >> _U64(27))) * _M2
        return z ^ (z >> _U64(31))


def stream_key(seed, stream=0):
    """Host-side key for one random stream: sm64(sm64(seed) ^ stream)."""
    return sm64(sm64(_U64(seed)) ^ _U64(stream))


def uniform16(key, idx):
    """u in [0,1) with 16 random bits, float32.  One hash serves four consecutive ids: h = sm64(key ^ (idx >> 2))
    and id idx reads the 16-bit field idx & 3 of h, most significant field first; u = field * 2^-16."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = sm64(_U64(key) ^ (idx >> _U64(2)))
    shift = _U64(48) - _U64(16) * (idx & _U64(3))
    return ((h >> shift) & _U64(0xFFFF)).astype(np.float32) * np.float32(2.0 ** -16)


def keep_mask(key, idx, keep_prob):
    """Dropout predicate of gcn/layers.py:21-23: floor(keep_prob + u) == 1, in float32."""
    u = uniform16(key, idx)
    return np.floor(np.float32(keep_prob) + u) >= np.float32(1.0)


def synthetic_dense(n, f, seed=7):
    """x[i,f] = (int32(h >> 32) >> 8) * 2^-23 in [-1, 1), float32, h = sm64(key ^ (i*F+f))."""
    key = stream_key(seed)
    idx = np.arange(n * f, dtype=np.uint64)
    h = sm64(key ^ idx)
    hi = (h >> _U64(32)).astype(np.uint32).view(np.int32)
    return ((hi >> 8).astype(np.float32) * np.float32(2.0 ** -23)).reshape(n, f)


# --------------------------------------------------------------------------- #
# R-MAT generator (the reference has none; integer-only so CPU == GPU)
# --------------------------------------------------------------------------- #
RMAT_A, RMAT_B, RMAT_C = 0.57, 0.19, 0.19


def rmat_thresholds():
    ta = int(RMAT_A * 2 ** 32)
    tb = int((RMAT_A + RMAT_B) * 2 ** 32)
    tc = int((RMAT_A + RMAT_B + RMAT_C) * 2 ** 32)
    return ta, tb, tc


def rmat_pairs(scale, edge_factor=16, seed=42):
    """The raw generated (u, v) pairs, int64 arrays of length edge_factor * 2^scale."""
    m = edge_factor << scale
    key = stream_key(seed)
    ta, tb, tc = rmat_thresholds()
    e = np.arange(m, dtype=np.uint64)
    u = np.zeros(m, dtype=np.int64)
    v = np.zeros(m, dtype=np.int64)
    for level in range(scale):
        r = (sm64(key ^ (e * _U64(scale) + _U64(level))) >> _U64(32)).astype(np.int64)
        q = (r >= ta).astype(np.int64) + (r >= tb) + (r >= tc)     # quadrant 0..3
        u = (u << 1) | (q >> 1)
        v = (v << 1) | (q & 1)
    return u, v


def scramble_id(v, bits, key):
    """Fixed bijection on [0, 2^bits): odd multiplications and xor-shifts modulo 2^bits."""
    mask = _U64((1 << bits) - 1)
    sh = _U64((bits + 1) // 2)
    with np.errstate(over='ignore'):
        x = (np.asarray(v, dtype=np.uint64) + _U64(key)) & mask
        x = (x * _GOLD) & mask
        x = x ^ (x >> sh)
        x = (x * _M1) & mask
        x = x ^ (x >> sh)
    return x


def rmat_csr(scale, edge_factor=16, seed=42, self_loops=False, scramble=True):
    """Simple undirected graph: scramble ids (unless ``scramble=False``), drop self-loops, symmetrise,
    sort, dedupe -> CSR structure; ``self_loops`` adds every (i, i) (the structure of A + I)."""
    u, v = rmat_pairs(scale, edge_factor, seed)
    keep = u != v
    u, v = u[keep], v[keep]
    if scramble:
        label_key = stream_key(seed, 1)
        u = scramble_id(u, scale, label_key).astype(np.int64)
        v = scramble_id(v, scale, label_key).astype(np.int64)
    keys = np.concatenate([(u << 32) | v, (v << 32) | u])
    n = 1 << scale
    if self_loops:
        d = np.arange(n, dtype=np.int64)
        keys = np.concatenate([keys, (d << 32) | d])
    keys = np.unique(keys)
    rows = (keys >> 32).astype(np.int64)
    cols = (keys & 0xFFFFFFFF).astype(np.int32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr), cols


def degree_order(indptr):
    """Vertices by descending row length, ties by ascending id: (new_id, old_id) int32."""
    deg = np.diff(np.asarray(indptr, dtype=np.int64))
    old_id = np.lexsort((np.arange(deg.size), -deg)).astype(np.int32)
    new_id = np.empty_like(old_id)
    new_id[old_id] = np.arange(deg.size, dtype=np.int32)
    return new_id, old_id


def hub_order(indptr, n_hubs):
    """Hub-first order: the n_hubs vertices of highest degree first (descending degree, ties by ascending id),
    every other vertex after them in its old order.  Returns (new_id, old_id) like degree_order."""
    by_degree, sorted_ids = degree_order(indptr)
    n = by_degree.size
    n_hubs = min(int(n_hubs), n)
    is_hub = by_degree < n_hubs
    old_id = np.concatenate([sorted_ids[:n_hubs], np.nonzero(~is_hub)[0]]).astype(np.int32)
    new_id = np.empty(n, dtype=np.int32)
    new_id[old_id] = np.arange(n, dtype=np.int32)
    return new_id, old_id


def csr_relabel(n_rows, indptr, indices, vals, new_row, new_col):
    """CSR of the matrix with entry (r, c) moved to (new_row[r], new_col[c]), columns ascending."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    r = np.asarray(new_row, dtype=np.int64)[rows]
    c = np.asarray(new_col, dtype=np.int64)[np.asarray(indices, dtype=np.int64)]
    order = np.lexsort((c, r))
    out_ip = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(out_ip, r + 1, 1)
    return np.cumsum(out_ip), c[order].astype(np.int32), (None if vals is None else np.asarray(vals)[order])


# --------------------------------------------------------------------------- #
# graph preprocessing: gcn/utils.py
# --------------------------------------------------------------------------- #
def coo_to_csr_sum(n_rows, n_cols, rows, cols, vals):
    """COO -> CSR with duplicates summed and columns ascending (what scipy's
    ``adj + sp.eye(n)`` / ``tocsr`` produce at gcn/utils.py:321)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    key = rows * n_cols + cols
    first = np.ones(key.size, dtype=bool)
    first[1:] = key[1:] != key[:-1]
    seg = np.cumsum(first) - 1
    out_vals = np.zeros(int(first.sum()), dtype=np.float64)
    np.add.at(out_vals, seg, vals)
    rows, cols = rows[first], cols[first]
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr), cols.astype(np.int32), out_vals


def add_self_loops(n, rows, cols, vals):
    """adj + sp.eye(n): ADDS 1.0 to every diagonal entry (gcn/utils.py:321)."""
    d = np.arange(n, dtype=np.int64)
    return (np.concatenate([rows, d]), np.concatenate([cols, d]),
            np.concatenate([np.asarray(vals, dtype=np.float64), np.ones(n)]))


def inv_power(rowsum, p):
    """np.power(rowsum, p) with inf -> 0 (gcn/utils.py:305-306, 311-312)."""
    with np.errstate(divide='ignore'):
        d = np.power(np.asarray(rowsum, dtype=np.float64), p)
    d[np.isinf(d)] = 0.0
    return d


def normalize_adj(n, indptr, indices, vals, type='sym'):
    """gcn/utils.py:297-315 on a CSR matrix.  Returns (rows, cols, values float64)
    in the element order the reference returns.

    'sym': result[i,j] = (a[j,i] * d[i]) * d[j], d = rowsum^-0.5, emitted in
    column-major order (CSR -> transpose -> CSC -> tocoo, utils.py:308).
    'rw' : result[i,j] = dinv[i] * a[i,j], emitted row-major (utils.py:309-315).
    """
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    cols = np.asarray(indices, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    rowsum = np.zeros(n, dtype=np.float64)
    # row sums in storage order (values are small integers on this path: exact)
    np.add.at(rowsum, rows, vals)
    if type == 'sym':
        d = inv_power(rowsum, -0.5)
        # entry stored at (rows=r, cols=c) of `a` lands at result[c, r]
        out_r, out_c = cols, rows
        out_v = (vals * d[out_r]) * d[out_c]
        # column-major order of the result == (out_c, out_r) ascending == CSR order of `a`
        return out_r, out_c, out_v
    elif type == 'rw':
        dinv = inv_power(rowsum, -1.0)
        return rows, cols, dinv[rows] * vals
    raise ValueError('unknown normalisation ' + str(type))


def preprocess_adj(n, rows, cols, vals=None, type='sym', loop=True):
    """gcn/utils.py:318-323.  Input: COO of ``adj``.  Output: the
    ``(indices [nnz,2] int32, values float64, dense_shape int64[2])`` triple of
    ``sparse_to_tuple`` (utils.py:258-275)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.ones(rows.size) if vals is None else np.asarray(vals, dtype=np.float64)
    if loop:
        rows, cols, vals = add_self_loops(n, rows, cols, vals)
    indptr, indices, data = coo_to_csr_sum(n, n, rows, cols, vals)
    r, c, v = normalize_adj(n, indptr, indices, data, type=type)
    coords = np.stack([r, c], axis=1).astype(np.int32)
    return coords, v, np.array([n, n], dtype=np.int64)


def triple_to_csr(coords, values, shape, dtype=np.float32, sort_columns=True):
    """Row-major CSR of a ``sparse_to_tuple`` triple; values cast the way the
    float32 sparse placeholder does at train.py:145-146.  ``sort_columns=False``
    keeps the triple's own order inside every row (stable)."""
    coords = np.asarray(coords)
    r = coords[:, 0].astype(np.int64)
    c = coords[:, 1].astype(np.int64)
    if sort_columns:
        order = np.lexsort((c, r))
    else:
        order = np.argsort(r, kind='stable')
    n = int(shape[0])
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    return np.cumsum(indptr), c[order].astype(np.int32), np.asarray(values)[order].astype(dtype), order


def preprocess_features_bow(indptr, indices, data, n_rows):
    """gcn/utils.py:279-285: r_inv = rowsum^-1 (inf -> 0); features = diag(r_inv) . features."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    m = sp.csr_matrix((data, indices, indptr), shape=(n_rows, int(np.max(indices)) + 1 if len(indices) else 1))
    rowsum = np.array(m.sum(1))
    with np.errstate(divide='ignore'):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    return (r_inv[rows] * data)


# --------------------------------------------------------------------------- #
# sparse x dense product: the serial loop tf.sparse_tensor_dense_matmul runs
# --------------------------------------------------------------------------- #
_clib = None


def _load_clib():
    global _clib
    if _clib is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, 'liboracle.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle/liboracle.so is not built: run `make -C oracle`')
        _clib = ctypes.CDLL(path)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        _clib.oracle_spmm_csr_f32.argtypes = [i64, i64, vp, vp, vp, vp, i64, vp, i64, ci, ci]
        _clib.oracle_spmm_csr_f32.restype = None
        _clib.oracle_spmm_coo_f32.argtypes = [i64, i64, i64, vp, vp, vp, vp, i64, vp, i64]
        _clib.oracle_spmm_coo_f32.restype = None
        _clib.oracle_num_threads.restype = ci
    return _clib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def spmm_csr(indptr, indices, vals, b, relu=False, threads=1, out=None):
    """C[i,:] = sum_k vals[k] * B[indices[k],:] for k in row i, float32, each
    output row accumulated sequentially in storage order with a separate multiply
    and add -- the order the nnz loop behind gcn/layers.py:31 produces for
    row-grouped indices.  Runs in C (liboracle.so)."""
    lib = _load_clib()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    n = indptr.size - 1
    f = b.shape[1]
    if out is None:
        out = np.empty((n, f), dtype=np.float32)
    lib.oracle_spmm_csr_f32(n, f, _ptr(indptr), _ptr(indices), _ptr(vals), _ptr(b), b.shape[1],
                            _ptr(out), out.shape[1], int(bool(relu)), int(threads))
    return out


def spmm_coo(n_rows, coords, vals, b):
    """The literal loop: out = 0; for every nnz in the given order
    out[row,:] += val * b[col,:] (what gcn/layers.py:31 executes on a triple)."""
    lib = _load_clib()
    coords = np.asarray(coords)
    r = np.ascontiguousarray(coords[:, 0], dtype=np.int32)
    c = np.ascontiguousarray(coords[:, 1], dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((n_rows, b.shape[1]), dtype=np.float32)
    lib.oracle_spmm_coo_f32(n_rows, r.size, b.shape[1], _ptr(r), _ptr(c), _ptr(vals), _ptr(b), b.shape[1],
                            _ptr(out), out.shape[1])
    return out


def spmm_csr_numpy(indptr, indices, vals, b):
    """Pure-numpy twin of ``spmm_csr`` for small cases (no compiled library needed)."""
    n = len(indptr) - 1
    b = np.asarray(b, dtype=np.float32)
    out = np.zeros((n, b.shape[1]), dtype=np.float32)
    for i in range(n):
        acc = np.zeros(b.shape[1], dtype=np.float32)
        for k in range(int(indptr[i]), int(indptr[i + 1])):
            acc = acc + np.float32(vals[k]) * b[indices[k]]
        out[i] = acc
    return out


def spmm_csr_f64(indptr, indices, vals, b):
    n = len(indptr) - 1
    m = sp.csr_matrix((np.asarray(vals, dtype=np.float64), indices, indptr), shape=(n, b.shape[0]))
    return m @ np.asarray(b, dtype=np.float64)


def csr_transpose(n_rows, n_cols, indptr, indices, vals):
    """Stable transpose: inside every output row the source rows ascend.  Returns
    (t_indptr, t_indices, t_vals, perm) with perm[k] = source position of entry k."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    cols = np.asarray(indices, dtype=np.int64)
    perm = np.argsort(cols, kind='stable')
    t_indptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.add.at(t_indptr, cols + 1, 1)
    return np.cumsum(t_indptr), rows[perm].astype(np.int32), np.asarray(vals)[perm], perm


# --------------------------------------------------------------------------- #
# layers: gcn/layers.py
# --------------------------------------------------------------------------- #
def sparse_dropout_values(vals, keep_prob, key):
    """gcn/layers.py:19-25 on the value array of a sparse tensor with nnz entries
    in feed order.  Returns (kept mask, scaled values); dropped entries vanish
    from the product (sparse_retain), kept ones are multiplied by float32(1/keep)."""
    vals = np.asarray(vals, dtype=np.float32)
    keep = keep_mask(key, np.arange(vals.size), keep_prob)
    inv = np.float32(1.0) / np.float32(keep_prob)
    return keep, vals * inv


def dense_dropout(x, keep_prob, key, index_offset=0):
    """tf.nn.dropout(x, keep_prob) as called at gcn/layers.py:185 (TF 1.x:
    ``div(x, keep_prob) * floor(keep_prob + uniform)``).  ``index_offset``: x is a row block of a taller
    matrix and draws that matrix's decisions (first row * width)."""
    x = np.asarray(x, dtype=np.float32)
    keep = keep_mask(key, np.arange(x.size) + index_offset, keep_prob).reshape(x.shape)
    return (x / np.float32(keep_prob)) * keep.astype(np.float32), keep


def glorot(shape, rng):
    """gcn/inits.py:11-15: U(-r, r), r = sqrt(6 / (fan_in + fan_out)), float32."""
    r = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-r, r, size=shape).astype(np.float32)


class GcnOracle(object):
    """Two-or-more-layer GCN of gcn/models.py:43-67 with connection 'c'*L.

    ``adj``  : (indptr, indices, vals32) CSR of the support, columns ascending.
    ``feat`` : (indptr, indices, vals32) CSR of X in FEED order (train.py:118).
    Forward, loss, accuracy and the analytic backward of gcn/models.py:92-113 and
    gcn/metrics.py:85-101.
    """

    def __init__(self, adj, feat, n, weights, weight_decay=5e-4):
        self.adj, self.feat, self.n = adj, feat, n
        self.w = [np.asarray(w, dtype=np.float32) for w in weights]
        self.wd = np.float32(weight_decay)
        f_in = self.w[0].shape[0]
        self.feat_t = csr_transpose(n, f_in, *feat)       # for X'^T . G

    # -- forward ----------------------------------------------------------- #
    def forward(self, dropout=0.0, seed=0, stream_ids=None):
        """gcn/layers.py:178-214 per layer: dropout -> X.W -> A_hat. -> act (no bias).
        ``stream_ids[li]`` names the random stream layer li's dropout draws from."""
        keep = np.float32(1.0) - np.float32(dropout)
        if stream_ids is None:
            stream_ids = list(range(len(self.w)))
        cache = {'keep': keep}
        ip, ix, xv = self.feat
        # layer 1: sparse inputs (models.py:50, 61)
        if dropout:
            m, xs = sparse_dropout_values(xv, keep, stream_key(seed, stream_ids[0]))
            xs = np.where(m, xs, np.float32(0))
            # dropped entries are skipped, not added as zeros
            cache['xmask'] = m
        else:
            xs = xv
            cache['xmask'] = None
        cache['xs'] = xs
        p = self._feat_matmul(xs, cache['xmask'], self.w[0])
        z = spmm_csr(*self.adj, p)
        cache['z'] = [z]
        cache['hin'] = [None]
        cache['hmask'] = [None]
        h = z
        for li in range(1, len(self.w)):
            h = np.maximum(h, np.float32(0))              # act of the previous layer (models.py:17)
            if dropout:
                hd, hm = dense_dropout(h, keep, stream_key(seed, stream_ids[li]))
            else:
                hd, hm = h, None
            cache['hin'].append(hd)
            cache['hmask'].append(hm)
            p = hd @ self.w[li]                           # layers.py:33
            z = spmm_csr(*self.adj, p)                    # layers.py:202
            cache['z'].append(z)
            h = z
        self.cache = cache
        return h                                          # last act is identity (models.py:18)

    def _feat_matmul(self, xs, mask, w):
        ip, ix, _ = self.feat
        if mask is None:
            return spmm_csr(ip, ix, xs, w)
        # drop the masked entries from the structure, keeping feed order
        rows = np.repeat(np.arange(self.n), np.diff(ip))
        r2, c2, v2 = rows[mask], ix[mask], xs[mask]
        ip2 = np.zeros(self.n + 1, dtype=np.int64)
        np.add.at(ip2, r2 + 1, 1)
        return spmm_csr(np.cumsum(ip2), c2, v2, w)

    # -- metrics ----------------------------------------------------------- #
    @staticmethod
    def masked_softmax_cross_entropy(logits, labels, mask):
        """gcn/metrics.py:85-92."""
        logits = logits.astype(np.float32)
        zmax = logits.max(axis=1, keepdims=True)
        lse = np.log(np.exp(logits - zmax).sum(axis=1, dtype=np.float32)) + zmax[:, 0]
        ce = (labels.astype(np.float32) * (lse[:, None] - logits)).sum(axis=1, dtype=np.float32)
        m = mask.astype(np.float32)
        m = m / m.mean(dtype=np.float32)
        return np.float32((ce * m).mean(dtype=np.float32))

    @staticmethod
    def masked_accuracy(logits, labels, mask):
        """gcn/metrics.py:94-101."""
        correct = (np.argmax(logits, 1) == np.argmax(labels, 1)).astype(np.float32)
        m = mask.astype(np.float32)
        m = m / m.mean(dtype=np.float32)
        return np.float32((correct * m).mean(dtype=np.float32))

    def loss(self, logits, labels, mask):
        """gcn/models.py:92-99: wd * l2_loss(var) for every layer var + masked CE."""
        total = np.float32(0)
        for w in self.w:
            total += self.wd * np.float32((w.astype(np.float32) ** 2).sum(dtype=np.float32) / 2)
        return np.float32(total + self.masked_softmax_cross_entropy(logits, labels, mask))

    # -- backward ---------------------------------------------------------- #
    def backward(self, logits, labels, mask):
        """Gradients of ``loss`` w.r.t. every weight (what models.py:86 asks for)."""
        c = self.cache
        n = self.n
        lab = labels.astype(np.float32)
        m = mask.astype(np.float32)
        m = m / m.mean(dtype=np.float32)
        z = logits.astype(np.float32)
        e = np.exp(z - z.max(axis=1, keepdims=True))
        sm = e / e.sum(axis=1, keepdims=True, dtype=np.float32)
        # d/dz of sum_c lab_c * (lse - z_c) = softmax * sum(lab) - lab
        dz = (sm * lab.sum(axis=1, keepdims=True) - lab) * (m / np.float32(n))[:, None]
        grads = [None] * len(self.w)
        at = csr_transpose(n, n, *self.adj)[:3]
        for li in range(len(self.w) - 1, 0, -1):
            dp = spmm_csr(*at, dz)                        # A_hat^T . dZ
            grads[li] = c['hin'][li].T @ dp + self.wd * self.w[li]
            dh = dp @ self.w[li].T
            if c['hmask'][li] is not None:
                dh = (dh / c['keep']) * c['hmask'][li].astype(np.float32)
            dz = dh * (c['z'][li - 1] > 0)
        dp = spmm_csr(*at, dz)
        tp, ti, _tv, perm = self.feat_t
        xs_t = c['xs'][perm]
        if c['xmask'] is not None:
            mk = c['xmask'][perm]
            f_in = self.w[0].shape[0]
            rows = np.repeat(np.arange(f_in), np.diff(tp))
            r2, c2, v2 = rows[mk], ti[mk], xs_t[mk]
            tp2 = np.zeros(f_in + 1, dtype=np.int64)
            np.add.at(tp2, r2 + 1, 1)
            g0 = spmm_csr(np.cumsum(tp2), c2, v2, dp)
        else:
            g0 = spmm_csr(tp, ti, xs_t, dp)
        grads[0] = g0 + self.wd * self.w[0]
        return grads


class TfAdam(object):
    """tf.train.AdamOptimizer(lr) as built at gcn/models.py:35-38 (defaults
    beta1 0.9, beta2 0.999, epsilon 1e-8): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    w -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        lr_t = np.float32(self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t))
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m += np.float32(1 - self.b1) * (g - m)
            v += np.float32(1 - self.b2) * (g * g - v)
            p -= lr_t * m / (np.sqrt(v) + np.float32(self.eps))
