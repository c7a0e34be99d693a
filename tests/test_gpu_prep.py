"""GPU parity: graph preprocessing kernels against the golden vectors made by the
reference's own gcn/utils.py (preprocess_adj, utils.py:318-323) and the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _t(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_coo_to_csr_with_self_loops_matches_reference(graph_fx):
    import torch
    from gcn_amd import sparse
    name, fx = graph_fx
    n = int(fx['n'])
    rng = np.random.RandomState(0)
    order = rng.permutation(fx['adj_row'].size)          # the kernel must not depend on input order
    rows, cols = _t(fx['adj_row'][order]), _t(fx['adj_col'][order])
    indptr, indices, vals = sparse.coo_to_csr(n, n, rows, cols, add_self_loops=True)
    assert np.array_equal(indptr.cpu().numpy(), fx['ai_indptr'].astype(np.int64))
    assert np.array_equal(indices.cpu().numpy(), fx['ai_indices'])
    assert np.array_equal(vals.cpu().numpy(), fx['ai_data'])            # diagonal 2.0 where A had a self-loop
    deg = sparse.row_sums(indptr, vals)
    assert np.array_equal(deg.cpu().numpy(), fx['deg'])
    # binary structure-only variant
    indptr_b, indices_b, none = sparse.coo_to_csr(n, n, rows, cols, add_self_loops=True, binary=True)
    assert none is None
    assert torch.equal(indptr_b, indptr) and torch.equal(indices_b, indices)


def _ulp_diff(a, b):
    """Distance in float64 ulps between same-signed finite arrays."""
    return np.abs(a.view(np.int64) - b.view(np.int64))


def _correctly_rounded_inv_sqrt(x):
    """x^-0.5 evaluated in 80-bit extended precision and rounded once to float64."""
    return (np.longdouble(1) / np.sqrt(x.astype(np.longdouble))).astype(np.float64)


def test_normalize_adj_values(graph_fx):
    """D^-1/2 (A+I) D^-1/2 in float64 (gcn/utils.py:297-308).  numpy's vectorised pow(x, -0.5) is not
    correctly rounded on every host (here 26 of the first 400 integers are 1 ulp off), so float64
    agreement with the reference vectors is asked to 4 ulp (two factors, each up to 1 ulp off); after the cast to float32 that the
    float32 placeholder makes (train.py:145) every value must be identical."""
    from gcn_amd import sparse
    name, fx = graph_fx
    n = int(fx['n'])
    indptr, indices, vals = sparse.coo_to_csr(n, n, _t(fx['adj_row']), _t(fx['adj_col']), add_self_loops=True)
    deg = sparse.row_sums(indptr, vals)
    d = sparse.degree_scale(deg, 'sym').cpu().numpy()
    assert np.array_equal(d, _correctly_rounded_inv_sqrt(fx['deg']))
    with np.errstate(divide='ignore'):
        assert _ulp_diff(d, np.power(fx['deg'], -0.5)).max() <= 1
    v64, v32 = sparse.normalize_values(indptr, indices, vals, _t(d), 'sym')
    # the triple of the reference is column-major: entry k of the CSR of (A+I) is result[col, row]
    rows = np.repeat(np.arange(n), np.diff(fx['ai_indptr']))
    assert np.array_equal(np.stack([fx['ai_indices'], rows], 1), fx['hat_indices'])
    assert _ulp_diff(v64.cpu().numpy(), fx["hat_values"]).max() <= 4
    assert np.array_equal(v32.cpu().numpy(), fx['hat_values'].astype(np.float32))
    # fed the reference's own d, the product itself is bit-exact
    with np.errstate(divide='ignore'):
        d_ref = np.power(fx['deg'], -0.5)
    v64r, _ = sparse.normalize_values(indptr, indices, vals, _t(d_ref), 'sym')
    assert np.array_equal(v64r.cpu().numpy(), fx['hat_values'])


def _is_correctly_rounded_inv_sqrt(x, y):
    """Exact rational check that float64 y is the float64 nearest to x^-0.5."""
    from fractions import Fraction
    lo = (Fraction(float(np.nextafter(y, 0.0))) + Fraction(float(y))) / 2
    hi = (Fraction(float(np.nextafter(y, np.inf))) + Fraction(float(y))) / 2
    fx = Fraction(float(x))
    return fx * lo * lo <= 1 <= fx * hi * hi


def test_degree_scale_is_correctly_rounded():
    """x^-0.5 in float64 for every integer degree up to 2^20 and a band of huge ones."""
    from gcn_amd import sparse
    x = np.concatenate([np.arange(1, 1 << 20, dtype=np.float64),
                        np.arange(1 << 30, (1 << 30) + 4096, dtype=np.float64),
                        np.array([0.25, 0.5, 1.5, 2.5, 1e-300, 1e300])])
    d = sparse.degree_scale(_t(x), 'sym').cpu().numpy()
    assert _ulp_diff(d, np.power(x, -0.5)).max() <= 1
    # the extended-precision reference rounds twice, so it is itself off in about 1 value of 2000:
    # where the two differ, and on a random sample where they agree, decide with exact arithmetic
    approx = _correctly_rounded_inv_sqrt(x)
    differ = np.flatnonzero(d != approx)
    assert differ.size < x.size // 500
    sample = np.random.RandomState(0).choice(x.size, 3000, replace=False)
    for i in np.concatenate([differ, sample]):
        assert _is_correctly_rounded_inv_sqrt(x[i], d[i]), (x[i], d[i])
    # zero degree -> 0 (utils.py:306), both normalisations
    z = np.zeros(3)
    assert np.array_equal(sparse.degree_scale(_t(z), 'sym').cpu().numpy(), z)
    assert np.array_equal(sparse.degree_scale(_t(z), 'rw').cpu().numpy(), z)
    dinv = sparse.degree_scale(_t(x), 'rw').cpu().numpy()
    assert np.array_equal(dinv, 1.0 / x)
    assert _ulp_diff(dinv, np.power(x, -1.0)).max() <= 1


def test_coo_to_csr_sums_duplicates_and_handles_empty_rows(oracle):
    from gcn_amd import sparse
    rng = np.random.RandomState(1)
    n_rows, n_cols = 1000, 700
    rows = rng.randint(0, n_rows, 20000)
    rows[rows % 7 == 0] = 3                       # many empty rows, one crowded row
    cols = rng.randint(0, 50, 20000)              # plenty of duplicates
    vals = rng.randint(1, 5, 20000).astype(np.float64)
    ip, ix, v = sparse.coo_to_csr(n_rows, n_cols, _t(rows), _t(cols), _t(vals))
    rip, rix, rv = oracle.coo_to_csr_sum(n_rows, n_cols, rows, cols, vals)
    assert np.array_equal(ip.cpu().numpy(), rip)
    assert np.array_equal(ix.cpu().numpy(), rix)
    assert np.array_equal(v.cpu().numpy(), rv)
    # no values given: duplicates count
    ip2, ix2, v2 = sparse.coo_to_csr(n_rows, n_cols, _t(rows), _t(cols))
    _, _, rv2 = oracle.coo_to_csr_sum(n_rows, n_cols, rows, cols, np.ones(rows.size))
    assert np.array_equal(v2.cpu().numpy(), rv2)
    # empty input
    import torch
    e = torch.empty(0, dtype=torch.int32, device='cuda')
    ip3, ix3, v3 = sparse.coo_to_csr(5, 5, e, e)
    assert ip3.cpu().tolist() == [0] * 6 and ix3.numel() == 0
    ip4, ix4, v4 = sparse.coo_to_csr(5, 5, e, e, add_self_loops=True)
    assert ip4.cpu().tolist() == [0, 1, 2, 3, 4, 5] and ix4.cpu().tolist() == [0, 1, 2, 3, 4]


def test_rows_to_indptr_and_transpose(dataset_fx, oracle):
    import torch
    from gcn_amd import sparse
    name, fx = dataset_fx
    n, f_in = (int(v) for v in fx['feat_shape'])
    coords = fx['feat_coords']
    indptr = sparse.rows_to_indptr(n, _t(coords[:, 0]))
    assert np.array_equal(indptr.cpu().numpy(), fx['feat_indptr'].astype(np.int64))
    x = sparse.DeviceCSR(n, f_in, indptr, _t(coords[:, 1]), _t(fx['feat_values'], torch.float32))
    xt = x.transpose()
    tp, ti, tv, perm = oracle.csr_transpose(n, f_in, fx['feat_indptr'], coords[:, 1],
                                            fx['feat_values'].astype(np.float32))
    assert np.array_equal(xt.indptr.cpu().numpy(), tp)
    assert np.array_equal(xt.indices.cpu().numpy(), ti)
    assert np.array_equal(xt.values.cpu().numpy(), tv)
    assert np.array_equal(xt.perm.cpu().numpy(), perm)
    with pytest.raises(ValueError):
        sparse.rows_to_indptr(n, _t(coords[::-1, 0].copy()))


def test_hat_is_symmetric_so_transpose_is_identity(graph_fx):
    import torch
    from gcn_amd import sparse
    name, fx = graph_fx
    n = int(fx['n'])
    a = sparse.DeviceCSR(n, n, _t(fx['ai_indptr'], torch.int64), _t(fx['ai_indices']),
                         _t(fx['hat_values'], torch.float32))
    at = a.transpose()
    assert torch.equal(at.indptr, a.indptr) and torch.equal(at.indices, a.indices)
    assert torch.equal(at.values, a.values)


def test_check_indices_raises_like_the_reference_kernel():
    import torch
    from gcn_amd import sparse
    indptr = torch.tensor([0, 2, 3], dtype=torch.int64, device='cuda')
    good = sparse.DeviceCSR(2, 3, indptr, torch.tensor([0, 2, 1], dtype=torch.int32, device='cuda'),
                            torch.ones(3, device='cuda'))
    good.check_indices()
    bad = sparse.DeviceCSR(2, 3, indptr, torch.tensor([0, 3, 1], dtype=torch.int32, device='cuda'),
                           torch.ones(3, device='cuda'))
    with pytest.raises(ValueError):
        bad.check_indices()


@pytest.mark.parametrize('scale,edge_factor', [(8, 16), (12, 16), (14, 4)])
@pytest.mark.parametrize('self_loops', [False, True])
def test_rmat_matches_oracle_bit_for_bit(oracle, scale, edge_factor, self_loops):
    from gcn_amd import sparse
    indptr, indices = sparse.rmat_csr(scale, edge_factor, seed=42, self_loops=self_loops)
    rip, rix = oracle.rmat_csr(scale, edge_factor, seed=42, self_loops=self_loops)
    assert np.array_equal(indptr.cpu().numpy(), rip)
    assert np.array_equal(indices.cpu().numpy(), rix)
    # simple undirected graph: symmetric, strictly ascending columns, diagonal only when asked for
    ip, ix = indptr.cpu().numpy(), indices.cpu().numpy()
    rows = np.repeat(np.arange(1 << scale), np.diff(ip))
    assert int((rows == ix).sum()) == ((1 << scale) if self_loops else 0)
    key = rows.astype(np.int64) << 32 | ix
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.sort(ix.astype(np.int64) << 32 | rows), key)


def test_synthetic_dense_matches_oracle(oracle):
    from gcn_amd import sparse
    x = sparse.synthetic_dense(1000, 37, seed=7)
    ref = oracle.synthetic_dense(1000, 37, seed=7)
    assert np.array_equal(x.cpu().numpy(), ref)
    assert ref.min() >= -1.0 and ref.max() < 1.0


def test_out_of_range_coo_indices_are_rejected_before_any_write():
    """A row or column index outside the shape (-1, n_rows, n_cols) is a ValueError, as scipy's coo_matrix
    and TensorFlow's sparse kernels make it, and the output buffers are left as they were."""
    import ctypes
    import torch
    from gcn_amd import _lib, sparse
    n = 6
    good_r = torch.tensor([0, 1, 2, 5], dtype=torch.int32, device='cuda')
    good_c = torch.tensor([1, 0, 3, 5], dtype=torch.int32, device='cuda')
    for bad_r, bad_c in ([-1, 0], [n, 0], [0, -1], [0, n], [2 ** 31 - 1, 0]):
        r = torch.cat([good_r, torch.tensor([bad_r], dtype=torch.int32, device='cuda')])
        c = torch.cat([good_c, torch.tensor([bad_c], dtype=torch.int32, device='cuda')])
        with pytest.raises(ValueError, match='outside'):
            sparse.coo_to_csr(n, n, r, c, add_self_loops=True)
        with pytest.raises(ValueError, match='outside'):
            sparse.coo_to_csr(n, n, r, c, binary=True)
    # through the C ABI with sentinel-filled outputs: nothing is written
    r = torch.tensor([0, 1, n], dtype=torch.int32, device='cuda')
    c = torch.tensor([0, 1, 2], dtype=torch.int32, device='cuda')
    cap, ws_bytes = ctypes.c_int64(), ctypes.c_size_t()
    _lib.call('gcn_coo_to_csr_bound', n, 3, _lib.GCN_BINARY, ctypes.byref(cap), ctypes.byref(ws_bytes))
    indptr = torch.full((n + 1 + 64,), -7, dtype=torch.int64, device='cuda')
    indices = torch.full((cap.value + 64,), -7, dtype=torch.int32, device='cuda')
    nnz_out = torch.full((1,), -7, dtype=torch.int64, device='cuda')
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device='cuda')
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    status = _lib.load().gcn_coo_to_csr(n, n, 3, r.data_ptr(), c.data_ptr(), None, _lib.GCN_BINARY, indptr.data_ptr(),
                                        indices.data_ptr(), None, nnz_out.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    torch.cuda.synchronize()
    assert status == _lib.GCN_ERR_INDEX_RANGE
    assert bool((indptr == -7).all()) and bool((indices == -7).all()) and int(nnz_out) == -7
    # gcn_rows_to_indptr: the same check on its row array
    out = torch.full((n + 1 + 64,), -7, dtype=torch.int64, device='cuda')
    for rows in ([0, 1, n], [-1, 0, 1]):
        rr = torch.tensor(rows, dtype=torch.int32, device='cuda')
        status = _lib.load().gcn_rows_to_indptr(n, 3, rr.data_ptr(), out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert status == _lib.GCN_ERR_INDEX_RANGE and bool((out == -7).all())
    # the valid part still converts
    ip, ix, _ = sparse.coo_to_csr(n, n, good_r, good_c, binary=True)
    assert ip.cpu().tolist() == [0, 1, 2, 3, 3, 3, 4] and ix.cpu().tolist() == [1, 0, 3, 5]


def test_degree_order_and_relabel_against_the_oracle(graph_fx, oracle):
    """Degree-descending vertex order and the relabelled CSR (integer work: bit-exact), then the product on
    the relabelled A_hat against the original one with B and C permuted the same way."""
    import torch
    from gcn_amd import sparse
    name, fx = graph_fx
    n = int(fx['n'])
    ip, ix, data = fx['ai_indptr'].astype(np.int64), fx['ai_indices'], fx['ai_data']
    d_ip, d_ix = _t(ip, torch.int64), _t(ix)
    new_id, old_id = sparse.degree_order(d_ip)
    ref_new, ref_old = oracle.degree_order(ip)
    assert np.array_equal(new_id.cpu().numpy(), ref_new) and np.array_equal(old_id.cpu().numpy(), ref_old)
    r_ip, r_ix, r_v = sparse.csr_relabel(n, n, d_ip, d_ix, _t(data), new_id, new_id)
    o_ip, o_ix, o_v = oracle.csr_relabel(n, ip, ix, data, ref_new, ref_new)
    assert np.array_equal(r_ip.cpu().numpy(), o_ip) and np.array_equal(r_ix.cpu().numpy(), o_ix)
    assert np.array_equal(r_v.cpu().numpy(), o_v)
    lens = np.diff(o_ip)
    assert np.all(lens[:-1] >= lens[1:])                                  # hub rows first
    # structure only, and a rectangular relabel of the columns alone
    s_ip, s_ix, s_v = sparse.csr_relabel(n, n, d_ip, d_ix, None, None, new_id)
    c_ip, c_ix, _ = oracle.csr_relabel(n, ip, ix, None, np.arange(n), ref_new)
    assert s_v is None and np.array_equal(s_ip.cpu().numpy(), c_ip) and np.array_equal(s_ix.cpu().numpy(), c_ix)
    # the product: A_hat' . B' with B' = B[old_id] equals (A_hat . B)[old_id] up to the order of additions
    hat = fx['hat_values'].astype(np.float32)
    a = sparse.DeviceCSR(n, n, d_ip, d_ix, _t(hat))
    h_ip, h_ix, h_v = sparse.csr_relabel(n, n, d_ip, d_ix, _t(fx['hat_values']), new_id, new_id)
    a2 = sparse.DeviceCSR(n, n, h_ip, h_ix.contiguous(), h_v.to(torch.float32))
    b = sparse.synthetic_dense(n, 64, seed=8)
    c = sparse.spmm(a, b)
    old64 = old_id.to(torch.int64)
    b2 = torch.empty_like(b)
    sparse.gather_rows(b, old64, b2)
    c2 = sparse.spmm(a2, b2)
    back = torch.empty_like(c)
    sparse.gather_rows(c2, new_id.to(torch.int64), back)                # permuted back to the original ids
    assert float((back - c).abs().max()) <= 1e-5 * float(c.abs().max())
    # a label outside the shape is refused
    bad = new_id.clone()
    bad[3] = n
    with pytest.raises(ValueError, match='outside'):
        sparse.csr_relabel(n, n, d_ip, d_ix, None, bad, new_id)


def test_rmat_natural_order_and_hot_column_hint(oracle):
    """The generator without the id scramble equals the oracle's; the non-temporal hint on cold columns
    (gcn_spmm_plan_set_hot_columns) changes no bit of a wide product."""
    import torch
    from gcn_amd import sparse
    scale = 14
    ip, ix = sparse.rmat_csr(scale, 16, seed=42, self_loops=True, scramble=False)
    o_ip, o_ix = oracle.rmat_csr(scale, 16, seed=42, self_loops=True, scramble=False)
    assert np.array_equal(ip.cpu().numpy(), o_ip) and np.array_equal(ix.cpu().numpy(), o_ix)
    n = 1 << scale
    lens = np.diff(o_ip)
    assert lens[0] == lens.max()                                        # vertex 0 is the largest hub
    d = sparse.degree_scale(sparse.row_sums(ip, None), 'sym')
    v = sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, ip, ix, v)
    for f in (256, 512):
        b = sparse.synthetic_dense(n, f, seed=2)
        plain = sparse.spmm(a, b)
        for hot in (1, 64, n // 2):
            a.plan.set_hot_columns(hot)
            assert torch.equal(sparse.spmm(a, b), plain)
        a.plan.set_hot_columns(0)


@pytest.mark.parametrize('n_hubs', [0, 1, 37, 4096, 1 << 20])
def test_hub_first_order_against_the_oracle(oracle, n_hubs):
    """gcn_hub_order: the n_hubs vertices of highest degree first, the others in their old order."""
    import torch
    from gcn_amd import sparse
    ip, ix = sparse.rmat_csr(14, 16, seed=42, self_loops=True)
    new_id, old_id = sparse.hub_order(ip, n_hubs)
    want_new, want_old = oracle.hub_order(ip.cpu().numpy(), n_hubs)
    assert np.array_equal(new_id.cpu().numpy(), want_new) and np.array_equal(old_id.cpu().numpy(), want_old)
    n = ip.numel() - 1
    assert torch.equal(torch.sort(new_id.to(torch.int64)).values, torch.arange(n, device='cuda'))
    k = min(n_hubs, n)
    rest = old_id[k:].to(torch.int64)
    assert bool((rest[1:] > rest[:-1]).all())                       # the others keep their order
    deg = (ip[1:] - ip[:-1])
    if 0 < k < n:
        assert int(deg[old_id[:k].to(torch.int64)].min()) >= int(deg[rest].max())
