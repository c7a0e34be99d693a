"""GPU parity of the sparse x dense products (gcn/layers.py:28-31, 198-202) through the C ABI.

Bar: bit-exact against the serial float32 loop of the oracle for every row of at
most ``split_nnz`` entries (all rows of Cora / CiteSeer / PubMed); for longer
rows max|x - ref64| <= 1e-5 * max|ref64| (float32 tolerance of the north star).
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _t(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _device_csr(n_rows, n_cols, indptr, indices, vals):
    import torch
    from gcn_amd import sparse
    return sparse.DeviceCSR(n_rows, n_cols, _t(indptr, torch.int64), _t(indices, torch.int32),
                            _t(vals, torch.float32))


def _hat(fx):
    n = int(fx['n'])
    return n, fx['ai_indptr'].astype(np.int64), fx['ai_indices'], fx['hat_values'].astype(np.float32)


@pytest.mark.parametrize('f', [16, 7, 6, 3, 1, 4, 32, 64, 100, 130, 256, 500, 512, 1028])
def test_hat_times_dense_bit_exact(graph_fx, oracle, f):
    from gcn_amd import sparse
    name, fx = graph_fx
    if name == 'pubmed_graph' and f in (100, 130, 1028):
        pytest.skip('covered on the smaller graphs')
    n, ip, ix, v = _hat(fx)
    a = _device_csr(n, n, ip, ix, v)
    assert a.plan.n_long_rows == 0                      # every row is summed in the serial order
    b = oracle.synthetic_dense(n, f, seed=11)
    ref = oracle.spmm_csr(ip, ix, v, b)
    out = sparse.spmm(a, _t(b)).cpu().numpy()
    assert np.array_equal(out, ref)
    out_relu = sparse.spmm(a, _t(b), relu=True).cpu().numpy()
    assert np.array_equal(out_relu, np.maximum(ref, 0))


def test_hat_times_dense_equals_the_committed_known_answers(graph_fx, oracle):
    """The HIP product against vectors committed with the repository (scipy's float32 CSR product = the
    serial loop of the reference's CPU kernel, oracle/make_known_answers.py), bit for bit."""
    from conftest import load_golden
    from gcn_amd import sparse
    name, fx = graph_fx
    known = load_golden('spmm_known_answers')
    n, ip, ix, v = _hat(fx)
    a = _device_csr(n, n, ip, ix, v)
    for key in known.files:
        if key.startswith(name + '_f'):
            f = int(key.split('_f')[1].split('_')[0])
            b = oracle.synthetic_dense(n, f, seed=int(known['seed']))
            assert np.array_equal(sparse.spmm(a, _t(b)).cpu().numpy(), known[key])


def test_the_triple_order_of_the_reference_gives_the_same_sums(graph_fx, oracle):
    """The literal nnz loop over the (column-major) triple of preprocess_adj equals the CSR row loop."""
    name, fx = graph_fx
    n, ip, ix, v = _hat(fx)
    b = oracle.synthetic_dense(n, 16, seed=5)
    assert np.array_equal(oracle.spmm_coo(n, fx['hat_indices'], fx['hat_values'].astype(np.float32), b),
                          oracle.spmm_csr(ip, ix, v, b))


def test_strided_operands(oracle):
    import torch
    from gcn_amd import sparse
    fx = load_golden('cora')
    n, ip, ix, v = _hat(fx)
    a = _device_csr(n, n, ip, ix, v)
    for f, ld in [(16, 24), (7, 8), (7, 9), (20, 33)]:
        b = oracle.synthetic_dense(n, f, seed=3)
        wide = torch.full((n, ld), float('nan'), device='cuda')
        wide[:, :f] = _t(b)
        out_wide = torch.full((n, ld), -7.0, device='cuda')
        sparse.spmm(a, wide[:, :f], out=out_wide[:, :f])
        ref = oracle.spmm_csr(ip, ix, v, b)
        assert np.array_equal(out_wide[:, :f].cpu().numpy(), ref)
        assert bool((out_wide[:, f:] == -7.0).all())        # nothing written outside the f columns


def _skewed_graph(rng, n, hubs, hub_deg, base_deg):
    rows, cols = [], []
    for r in range(n):
        d = hub_deg if r in hubs else rng.randint(0, base_deg + 1)
        c = np.sort(rng.choice(n, size=min(d, n), replace=False))
        rows.append(np.full(c.size, r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    ip = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ip, rows + 1, 1)
    return np.cumsum(ip), cols.astype(np.int32)


@pytest.mark.parametrize('f', [16, 256, 36, 5])
def test_long_rows_are_split_and_empty_rows_are_zero(oracle, f):
    from gcn_amd import sparse
    rng = np.random.RandomState(2)
    n = 6000
    hubs = {0, 17, 2999, 5999}
    ip, ix = _skewed_graph(rng, n, hubs, hub_deg=5000, base_deg=6)
    v = rng.uniform(-1, 1, ix.size).astype(np.float32)
    a = _device_csr(n, n, ip, ix, v)
    assert a.plan.n_long_rows == len(hubs)
    assert a.plan.n_chunks >= len(hubs) * -(-5000 // a.plan.split_nnz)      # chunks are at most the threshold long
    b = oracle.synthetic_dense(n, f, seed=9)
    out = sparse.spmm(a, _t(b)).cpu().numpy()
    ref32 = oracle.spmm_csr(ip, ix, v, b)
    ref64 = oracle.spmm_csr_f64(ip, ix, v, b)
    short = np.diff(ip) <= a.plan.split_nnz
    assert np.array_equal(out[short], ref32[short])
    assert np.all(out[np.diff(ip) == 0] == 0)
    assert np.abs(out - ref64).max() <= REL_TOL * np.abs(ref64).max()
    np.testing.assert_allclose(out, ref64, rtol=REL_TOL, atol=REL_TOL * np.abs(ref64).max())
    out_relu = sparse.spmm(a, _t(b), relu=True).cpu().numpy()
    assert np.array_equal(out_relu, np.maximum(out, 0))


def test_degenerate_shapes(oracle):
    import torch
    from gcn_amd import sparse
    # no nonzeros at all
    a = _device_csr(5, 4, np.zeros(6, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32))
    out = sparse.spmm(a, torch.ones(4, 8, device='cuda'))
    assert out.shape == (5, 8) and bool((out == 0).all())
    # zero-width dense operand
    out = sparse.spmm(a, torch.ones(4, 0, device='cuda'))
    assert out.shape == (5, 0)
    # zero rows
    a0 = _device_csr(0, 4, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32))
    assert sparse.spmm(a0, torch.ones(4, 8, device='cuda')).shape == (0, 8)
    # shape mismatch is a ValueError, as a bad feed is in the reference
    with pytest.raises(ValueError):
        sparse.spmm(a, torch.ones(3, 8, device='cuda'))
    # trailing empty rows and a single entry
    a1 = _device_csr(4, 4, np.array([0, 1, 1, 1, 1]), np.array([2], dtype=np.int32), np.array([0.5], dtype=np.float32))
    b = torch.arange(16, dtype=torch.float32, device='cuda').reshape(4, 4)
    out = sparse.spmm(a1, b).cpu().numpy()
    assert np.array_equal(out[0], 0.5 * b[2].cpu().numpy()) and np.all(out[1:] == 0)


def test_rmat_product_properties(oracle):
    """Scale-16 R-MAT, F=64: bit-exact short rows, tolerance on hubs, linearity."""
    import torch
    from gcn_amd import sparse
    scale = 16
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42)
    n = 1 << scale
    ones = torch.ones(indices.numel(), dtype=torch.float64, device='cuda')
    # A + I through the preprocessing kernels ...
    rows = torch.repeat_interleave(torch.arange(n, device='cuda'), indptr[1:] - indptr[:-1])
    ip, ix, vals = sparse.coo_to_csr(n, n, rows, indices, ones, add_self_loops=True)
    # ... equals the structure the generator emits directly
    ip_direct, ix_direct = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    assert torch.equal(ip, ip_direct) and torch.equal(ix, ix_direct) and bool((vals == 1).all())
    deg = sparse.row_sums(ip, vals)
    assert torch.equal(deg, sparse.row_sums(ip, None))
    d = sparse.degree_scale(deg, 'sym')
    _, v32 = sparse.normalize_values(ip, ix, vals, d, 'sym')
    assert torch.equal(v32, sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1])
    a = sparse.DeviceCSR(n, n, ip, ix, v32)
    assert a.plan.n_long_rows > 0
    b = sparse.synthetic_dense(n, 64, seed=7)
    out = sparse.spmm(a, b)
    ipn, ixn, vn, bn = ip.cpu().numpy(), ix.cpu().numpy(), v32.cpu().numpy(), b.cpu().numpy()
    ref32 = oracle.spmm_csr(ipn, ixn, vn, bn, threads=8)
    ref64 = oracle.spmm_csr_f64(ipn, ixn, vn, bn)
    o = out.cpu().numpy()
    short = np.diff(ipn) <= a.plan.split_nnz
    assert np.array_equal(o[short], ref32[short])
    assert np.abs(o - ref64).max() <= REL_TOL * np.abs(ref64).max()
    # linearity: A(2B) == 2 A B exactly (scaling by 2 is exact in float32)
    assert torch.equal(sparse.spmm(a, 2 * b), 2 * out)
    # symmetric operator: transpose is the same matrix
    at = a.transpose()
    assert torch.equal(at.indices, a.indices) and torch.equal(at.values, a.values)


@pytest.mark.parametrize('keep_prob', [0.5, 0.9, 1.0])
def test_sparse_dropout_product_bit_exact(dataset_fx, oracle, keep_prob):
    """X' . W0 with sparse_dropout fused (gcn/layers.py:19-25, 198-199), and X'^T . G."""
    import torch
    from gcn_amd import sparse
    name, fx = dataset_fx
    n, f_in = (int(v) for v in fx['feat_shape'])
    ip = fx['feat_indptr'].astype(np.int64)
    ix = fx['feat_coords'][:, 1]
    xv = fx['feat_values'].astype(np.float32)
    x = _device_csr(n, f_in, ip, ix, xv)
    w = oracle.synthetic_dense(f_in, 16, seed=21)
    seed, stream_id = 1234, 6
    key = oracle.stream_key(seed, stream_id)
    keep, scaled = oracle.sparse_dropout_values(xv, keep_prob, key)
    if keep_prob == 0.5:
        assert 0.45 < keep.mean() < 0.55
    rows = np.repeat(np.arange(n), np.diff(ip))
    ip2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ip2, rows[keep] + 1, 1)
    ref = oracle.spmm_csr(np.cumsum(ip2), ix[keep], scaled[keep], w)
    out = sparse.spmm_dropout(x, _t(w), keep_prob, seed, stream_id).cpu().numpy()
    assert np.array_equal(out, ref)
    # adjoint with the SAME mask: dW0 = X'^T . G, entry ids routed through perm
    g = oracle.synthetic_dense(n, 16, seed=22)
    xt = x.transpose()
    tp, ti, tv, perm = oracle.csr_transpose(n, f_in, ip, ix, scaled)
    kt = keep[perm]
    trows = np.repeat(np.arange(f_in), np.diff(tp))
    tp2 = np.zeros(f_in + 1, dtype=np.int64)
    np.add.at(tp2, trows[kt] + 1, 1)
    ref_t = oracle.spmm_csr(np.cumsum(tp2), ti[kt], tv[kt], g)
    out_t = sparse.spmm_dropout(xt, _t(g), keep_prob, seed, stream_id).cpu().numpy()
    # a feature used by more than split_nnz documents is a long row of X^T: summed chunk-wise
    short = np.diff(tp) <= xt.plan.split_nnz
    assert np.array_equal(out_t[short], ref_t[short])
    ref_t64 = oracle.spmm_csr_f64(np.cumsum(tp2), ti[kt], tv[kt], g)
    assert np.abs(out_t - ref_t64).max() <= REL_TOL * np.abs(ref_t64).max()
    # the plan a gradient product asks for: rows cut at 64 entries (parallelism on a small matrix);
    # rows up to 64 entries keep the serial order, every row stays within the tolerance
    x2 = _device_csr(n, f_in, ip, ix, xv)
    xg = x2.transpose(for_gradient=True)
    assert xg.plan.split_nnz == sparse.GRADIENT_SPLIT == 64 and xg.plan.n_long_rows > xt.plan.n_long_rows
    out_g = sparse.spmm_dropout(xg, _t(g), keep_prob, seed, stream_id).cpu().numpy()
    short = np.diff(tp) <= 64
    assert np.array_equal(out_g[short], ref_t[short])
    assert np.abs(out_g - ref_t64).max() <= REL_TOL * np.abs(ref_t64).max()
    assert np.array_equal(sparse.spmm_dropout(xg, _t(g), keep_prob, seed, stream_id).cpu().numpy(), out_g)   # deterministic


@pytest.mark.parametrize('f', [16, 256, 36, 5, 512])
@pytest.mark.parametrize('keep_prob', [0.5, 0.8])
def test_dropout_in_the_store_equals_the_separate_pass(oracle, f, keep_prob):
    """dropout(relu(A . B)) written by the product itself (the next layer's tf.nn.dropout,
    gcn/layers.py:185): the bits of the product followed by the oracle's dense dropout, for short rows,
    chunked hub rows, the element path (f = 5), a row block and a column range of a wider matrix."""
    import torch
    from gcn_amd import sparse
    rng = np.random.RandomState(4)
    n = 3000
    ip, ix = _skewed_graph(rng, n, {3, 1500}, hub_deg=2500, base_deg=6)
    v = rng.uniform(-1, 1, ix.size).astype(np.float32)
    a = _device_csr(n, n, ip, ix, v)
    b = _t(oracle.synthetic_dense(n, f, seed=10))
    plain = sparse.spmm(a, b, relu=True)
    row0 = 777                                              # this matrix is rows 777.. of a taller one
    od = sparse.OutDropout(keep_prob, 21, 5, index_offset=row0 * f)
    fused = sparse.spmm(a, b, relu=True, out_dropout=od)
    ref, _ = oracle.dense_dropout(plain.cpu().numpy(), keep_prob, oracle.stream_key(21, 5), index_offset=row0 * f)
    assert np.array_equal(fused.cpu().numpy(), ref)
    assert torch.equal(fused, sparse.dropout(plain, keep_prob, 21, 5, index_offset=row0 * f))
    assert torch.equal(fused > 0, (plain > 0) & (fused != 0))      # the dropped output is its own backward gate
    if f >= 16:                                             # a column range of the same output
        c0, c1 = 4, 4 + 8
        part = torch.full((n, 8), -3.0, device='cuda')
        sparse.spmm(a, b[:, c0:c1], relu=True, out=part, wide_split=f >= 128,
                    out_dropout=sparse.OutDropout(keep_prob, 21, 5, index_offset=row0 * f, pitch=f).columns(c0))
        assert torch.equal(part, fused[:, c0:c1])
