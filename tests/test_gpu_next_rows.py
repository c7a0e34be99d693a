"""GPU parity for the callers and data formats next to the hot path: bag-of-words row scaling
(gcn/utils.py:279-285), the sibling supports gcn_noloop / gcn_rw / gcn_unnorm (train.py:127-135),
repeated propagation A_hat^k X on the PubMed graph, and the absorption-probability approximation
(gcn/utils.py:559-571) with its residual fused into the product."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _t(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _adj(fx):
    n = int(fx['n'])
    return sp.csr_matrix((np.ones(fx['adj_row'].size), (fx['adj_row'], fx['adj_col'])), shape=(n, n))


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_bag_of_words_row_scaling(name):
    """Raw 0/1 features -> the reference's preprocessed values, bit for bit, in the matrix's own dtype."""
    import torch
    from gcn_amd import sparse
    fx = load_golden(name)
    raw = np.ones_like(fx['feat_data'])                        # Planetoid features are 0/1 before scaling
    out = sparse.row_normalize(_t(fx['feat_indptr'], torch.int64), _t(raw))
    assert out.dtype == (torch.float32 if fx['feat_data'].dtype == np.float32 else torch.float64)
    got = out.cpu().numpy()
    # the kernel returns the correctly rounded 1 / rowsum; numpy's vectorised pow(x, -1) that made the
    # reference values is 1 ulp off for some counts (float32 1/13 on this host), so: exact against
    # IEEE division, within 1 ulp of the reference vector
    counts = np.repeat(np.diff(fx['feat_indptr']), np.diff(fx['feat_indptr'])).astype(got.dtype)
    assert np.array_equal(got, np.ones_like(got) / counts)
    as_int = np.int32 if got.dtype == np.float32 else np.int64
    assert np.abs(got.view(as_int).astype(np.int64) - fx['feat_data'].view(as_int).astype(np.int64)).max() <= 1
    # an empty row stays empty and a zero row sum scales to 0 (utils.py:283)
    ip = _t(np.array([0, 0, 2, 3]), torch.int64)
    v = sparse.row_normalize(ip, _t(np.array([1.0, -1.0, 4.0], dtype=np.float32)))
    assert v.cpu().tolist() == [0.0, 0.0, 1.0]


def test_sibling_supports_on_the_gpu(graph_fx):
    from gcn_amd.utils import preprocess_adj
    name, fx = graph_fx
    n = int(fx['n'])
    adj = _adj(fx)
    # gcn_noloop: same normalisation without the identity
    s = preprocess_adj(adj, loop=False)
    assert np.array_equal(s.indices, fx['noloop_indices'])
    assert np.abs(s.values.view(np.int64) - fx['noloop_values'].view(np.int64)).max() <= 4
    assert s.csr.nnz == fx['noloop_values'].size
    # gcn_rw: D^-1 (A + I), row-major; the reference keeps scipy's unsorted order inside a row
    s = preprocess_adj(adj, type='rw')
    key_ref = fx['rw_indices'][:, 0].astype(np.int64) * n + fx['rw_indices'][:, 1]
    order = np.argsort(key_ref)
    assert np.array_equal(s.indices[:, 0].astype(np.int64) * n + s.indices[:, 1], key_ref[order])
    assert np.abs(s.values.view(np.int64) - fx['rw_values'][order].view(np.int64)).max() <= 1
    # every row of D^-1 (A+I) sums to 1
    from gcn_amd import sparse
    import torch
    ones = torch.ones((n, 4), device='cuda')
    assert float((sparse.spmm(s.csr, ones) - 1).abs().max()) < 1e-5
    # rw is not symmetric: its transpose is a different matrix, and (A^T)^T is A again
    t = s.csr.transpose()
    assert not torch.equal(t.values, s.csr.values)
    tt = t.transpose()
    assert torch.equal(tt.indices, s.csr.indices) and torch.equal(tt.values, s.csr.values)


def test_unnormalised_support_through_the_model_feed(oracle):
    """gcn_unnorm feeds sparse_to_tuple(adj.astype(float32)) directly (train.py:128)."""
    from gcn_amd import sparse
    from gcn_amd.utils import sparse_to_tuple
    fx = load_golden('cora')
    n = int(fx['n'])
    triple = sparse_to_tuple(_adj(fx).astype(np.float32))
    csr = sparse.triple_to_device_csr(triple)
    b = oracle.synthetic_dense(n, 16, seed=2)
    ip = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ip, fx['adj_row'].astype(np.int64) + 1, 1)
    ref = oracle.spmm_csr(np.cumsum(ip), fx['adj_col'], np.ones(fx['adj_col'].size, dtype=np.float32), b)
    assert np.array_equal(sparse.spmm(csr, _t(b)).cpu().numpy(), ref)


def test_repeated_propagation_on_pubmed_stays_bit_exact(oracle):
    """A_hat^k X, k = 1..64, X = 19717 x 500: every row has at most 172 entries, so each step is the
    serial sum and the GPU equals the float32 oracle after 64 steps; against float64 the error
    grows slowly with k."""
    import torch
    from gcn_amd import smoothing, sparse
    fx = load_golden('pubmed_graph')
    n = int(fx['n'])
    ip, ix, v = fx['ai_indptr'].astype(np.int64), fx['ai_indices'], fx['hat_values'].astype(np.float32)
    a = sparse.DeviceCSR(n, n, _t(ip), _t(ix), _t(v))
    x = oracle.synthetic_dense(n, 500, seed=7)
    ref32, ref64 = x.copy(), x.astype(np.float64)
    m64 = sp.csr_matrix((v.astype(np.float64), ix, ip), shape=(n, n))
    checkpoints = {1, 2, 4, 8, 16, 32, 64}
    dx = _t(x)
    for k in range(1, 65):
        ref32 = oracle.spmm_csr(ip, ix, v, ref32, threads=8)
        ref64 = m64 @ ref64
        if k in checkpoints:
            got = smoothing.propagate(a, dx, k).cpu().numpy()
            assert np.array_equal(got, ref32), k
            assert np.abs(got - ref64).max() <= REL_TOL * np.abs(ref64).max(), k
    assert torch.equal(smoothing.propagate(a, dx, 0), dx)


@pytest.mark.parametrize('name', ['cora'])
def test_ap_approximate_against_the_float64_recurrence(name):
    from gcn_amd import smoothing
    fx = load_golden(name)
    n = int(fx['n'])
    adj = _adj(fx)
    feats = sp.csr_matrix((fx['feat_data'], fx['feat_indices'], fx['feat_indptr']),
                          shape=tuple(int(v) for v in fx['feat_shape'])).toarray()[:, :64].astype(np.float64)
    alpha, k = 0.5, 10
    # the reference's recurrence in float64 (utils.py:560-570), l1 row normalisation written out
    m = sp.csr_matrix(adj + sp.eye(n))
    m = sp.diags(1.0 / np.asarray(m.sum(1)).ravel()) @ m / (alpha + 1)
    ref = np.zeros_like(feats)
    for _ in range(k):
        ref = m @ ref + feats
    ref *= alpha / (alpha + 1)
    got = smoothing.ap_approximate(adj, feats, alpha, k)
    assert np.abs(got - ref).max() <= REL_TOL * np.abs(ref).max()


def test_fused_addend_equals_product_then_add(oracle):
    import torch
    from gcn_amd import sparse
    rng = np.random.RandomState(4)
    for n, f, hub in [(3000, 16, 0), (3000, 36, 2500), (2000, 256, 1500), (500, 5, 0)]:
        rows, cols = [], []
        for r in range(n):
            d = hub if (hub and r % 997 == 0) else rng.randint(0, 8)
            c = np.sort(rng.choice(n, size=d, replace=False))
            rows.append(np.full(c.size, r))
            cols.append(c)
        rows, cols = np.concatenate(rows), np.concatenate(cols).astype(np.int32)
        ip = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ip, rows + 1, 1)
        ip = np.cumsum(ip)
        v = rng.uniform(-1, 1, cols.size).astype(np.float32)
        a = sparse.DeviceCSR(n, n, _t(ip), _t(cols), _t(v))
        b, d = oracle.synthetic_dense(n, f, seed=1), oracle.synthetic_dense(n, f, seed=2)
        plain = sparse.spmm(a, _t(b))
        fused = sparse.spmm_add(a, _t(b), _t(d))
        assert torch.equal(fused, plain + _t(d))
        fused_relu = sparse.spmm_add(a, _t(b), _t(d), relu=True)
        assert torch.equal(fused_relu, torch.relu(plain + _t(d)))


def test_graphed_propagation_replays_the_same_bits(oracle):
    """A_hat^k X on PubMed recorded as one HIP graph: equal to the step-by-step products, on a
    second input too (the replay reads the new contents of the recorded input buffer)."""
    import torch
    from gcn_amd import smoothing, sparse
    fx = load_golden('pubmed_graph')
    n = int(fx['n'])
    a = sparse.DeviceCSR(n, n, torch.as_tensor(fx['ai_indptr'].astype(np.int64)).cuda(),
                         torch.as_tensor(fx['ai_indices'].astype(np.int32)).cuda(),
                         torch.as_tensor(fx['hat_values'].astype(np.float32)).cuda())
    k, f = 8, 500
    graphed = smoothing.GraphedPropagation(a, f, k)
    for seed in (7, 8):
        x = sparse.synthetic_dense(n, f, seed=seed)
        assert torch.equal(graphed(x), smoothing.propagate(a, x, k))
    smooth = smoothing.GraphedPropagation(a, f, 3, addend=True)
    x = sparse.synthetic_dense(n, f, seed=9)
    cur = torch.zeros_like(x)
    for _ in range(3):
        cur = sparse.spmm_add(a, cur, x)
    assert torch.equal(smooth(x), cur)
