"""CPU: pin the oracle (oracle/gcn_oracle.py) against the vectors produced by the reference's own
gcn/utils.py (tests/golden/*.npz, made by oracle/make_fixtures.py)."""
import numpy as np
import scipy.sparse as sp

from conftest import load_golden


def test_preprocess_adj_matches_reference_bit_for_bit(graph_fx, oracle):
    """gcn/utils.py:318-323 -- indices (column-major triple), float64 values, dense_shape."""
    name, fx = graph_fx
    n = int(fx['n'])
    coords, vals, shape = oracle.preprocess_adj(n, fx['adj_row'], fx['adj_col'])
    assert coords.dtype == np.int32 and np.array_equal(coords, fx['hat_indices'])
    assert vals.dtype == np.float64 and np.array_equal(vals, fx['hat_values'])
    assert np.array_equal(shape, fx['hat_shape'])
    # the triple is sorted by column, then row
    assert np.all(np.diff(coords[:, 1].astype(np.int64) * n + coords[:, 0]) > 0)


def test_sibling_supports(graph_fx, oracle):
    """loop=False (train.py:131) and type='rw' (train.py:134)."""
    name, fx = graph_fx
    n = int(fx['n'])
    c, v, _ = oracle.preprocess_adj(n, fx['adj_row'], fx['adj_col'], loop=False)
    assert np.array_equal(c, fx['noloop_indices']) and np.array_equal(v, fx['noloop_values'])
    # 'rw' keeps scipy's storage order inside a row (an artefact of `adj + eye`): compare as sets
    c, v, _ = oracle.preprocess_adj(n, fx['adj_row'], fx['adj_col'], type='rw')
    ref_key = fx['rw_indices'][:, 0].astype(np.int64) * n + fx['rw_indices'][:, 1]
    order = np.argsort(ref_key)
    assert np.array_equal(c[:, 0].astype(np.int64) * n + c[:, 1], ref_key[order])
    assert np.array_equal(v, fx['rw_values'][order])


def test_self_loops_are_added_not_set(oracle):
    """CiteSeer has 124 nodes with a self-loop in A: their diagonal of A+I is 2 (utils.py:321)."""
    fx = load_golden('citeseer')
    n = int(fx['n'])
    diag_in_a = int((fx['adj_row'] == fx['adj_col']).sum())
    assert diag_in_a == 124
    rows = np.repeat(np.arange(n), np.diff(fx['ai_indptr']))
    diag = fx['ai_data'][rows == fx['ai_indices']]
    assert diag.size == n and int((diag == 2).sum()) == 124
    r, c, v = oracle.add_self_loops(n, fx['adj_row'].astype(np.int64), fx['adj_col'].astype(np.int64),
                                    np.ones(fx['adj_row'].size))
    ip, ix, data = oracle.coo_to_csr_sum(n, n, r, c, v)
    assert np.array_equal(ip, fx['ai_indptr']) and np.array_equal(ix, fx['ai_indices'])
    assert np.array_equal(data, fx['ai_data'])
    assert np.array_equal(np.bincount(rows, weights=fx['ai_data'], minlength=n), fx['deg'])


def test_csr_of_the_triple(graph_fx, oracle):
    name, fx = graph_fx
    ip, ix, v32, order = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    assert np.array_equal(ip, fx['ai_indptr']) and np.array_equal(ix, fx['ai_indices'])
    # A_hat is symmetric in float32 (so A_hat^T . G may reuse A_hat)
    n = int(fx['n'])
    m = sp.csr_matrix((v32, ix, ip), shape=(n, n))
    assert (m != m.T).nnz == 0


def test_serial_spmm_forms_agree(graph_fx, oracle):
    """C loop (CSR), literal COO loop over the reference's triple, scipy float32 CSR @, numpy loop."""
    name, fx = graph_fx
    n = int(fx['n'])
    ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    for f in (16, 7):
        b = oracle.synthetic_dense(n, f, seed=5)
        c = oracle.spmm_csr(ip, ix, v32, b)
        assert np.array_equal(c, oracle.spmm_coo(n, fx['hat_indices'], fx['hat_values'].astype(np.float32), b))
        assert np.array_equal(c, sp.csr_matrix((v32, ix, ip), shape=(n, n)) @ b)
        assert np.array_equal(c[:64], oracle.spmm_csr_numpy(ip[:65], ix, v32, b))
        assert np.array_equal(c, oracle.spmm_csr(ip, ix, v32, b, threads=4))
        c64 = oracle.spmm_csr_f64(ip, ix, v32, b)
        assert np.abs(c - c64).max() <= 1e-6 * np.abs(c64).max()


def test_oracle_reproduces_the_committed_known_answers(graph_fx, oracle):
    """A_hat . B for seeded B, as scipy's float32 CSR product (the serial loop of the reference's CPU
    kernel) computed it when the vectors were committed (oracle/make_known_answers.py): the C loop gives
    the same bits today."""
    name, fx = graph_fx
    known = load_golden('spmm_known_answers')
    n = int(fx['n'])
    ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    widths = [int(k.split('_f')[1].split('_')[0]) for k in known.files if k.startswith(name + '_f') and k.endswith('_f32')]
    assert len(widths) >= 1
    for f in widths:
        b = oracle.synthetic_dense(n, f, seed=int(known['seed']))
        c = oracle.spmm_csr(ip, ix, v32, b)
        assert np.array_equal(c, known['%s_f%d_f32' % (name, f)])


def test_feature_triple_order(dataset_fx, oracle):
    """sparse_to_tuple(features) (train.py:118): row-grouped, order inside a row as stored."""
    name, fx = dataset_fx
    coords = fx['feat_coords']
    assert np.all(np.diff(coords[:, 0]) >= 0)
    assert np.array_equal(coords[:, 1], fx['feat_indices'])
    assert np.array_equal(fx['feat_values'], fx['feat_data'])
    n = int(fx['feat_shape'][0])
    ip = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ip, coords[:, 0].astype(np.int64) + 1, 1)
    assert np.array_equal(np.cumsum(ip), fx['feat_indptr'])
    # bag-of-words rows sum to 1 (or 0 for CiteSeer's padded isolated nodes, utils.py:127-136)
    sums = np.add.reduceat(np.append(fx['feat_values'], 0), fx['feat_indptr'][:-1].astype(np.int64))
    sums[np.diff(fx['feat_indptr']) == 0] = 0
    assert np.all((np.abs(sums - 1) < 1e-5) | (sums == 0))


def test_dropout_rule_and_counter_rng(oracle):
    """keep iff floor(keep_prob + u) == 1 (layers.py:21-23); u has 16 bits, four consecutive ids share one
    hash, and the stream is reproducible."""
    key = oracle.stream_key(1234, 6)
    u = oracle.uniform16(key, np.arange(1 << 16))
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1
    assert np.array_equal(u, oracle.uniform16(key, np.arange(1 << 16)))
    assert np.all(u * 2 ** 16 == np.round(u * 2 ** 16))
    h = oracle.sm64(np.uint64(key) ^ np.arange(1 << 14, dtype=np.uint64))            # id >> 2 of the ids above
    fields = np.stack([(h >> np.uint64(48 - 16 * j)) & np.uint64(0xFFFF) for j in range(4)], axis=1).reshape(-1)
    assert np.array_equal((u * 2 ** 16).astype(np.uint64), fields)
    assert abs(u.mean() - 0.5) < 0.01
    for keep_prob in (0.5, 0.9, 1.0):
        keep = oracle.keep_mask(key, np.arange(1 << 16), keep_prob)
        assert np.array_equal(keep, u >= np.float32(1) - np.float32(keep_prob)) or keep_prob == 0.9
        assert abs(keep.mean() - keep_prob) < 0.01
    assert oracle.stream_key(1234, 6) != oracle.stream_key(1234, 7)
    x = np.ones((100, 16), dtype=np.float32)
    y, m = oracle.dense_dropout(x, 0.5, key)
    assert set(np.unique(y)) == {0.0, 2.0} and np.array_equal(y > 0, m)


def test_rmat_generator_properties(oracle):
    ip, ix = oracle.rmat_csr(10, 16, seed=42)
    n = 1 << 10
    rows = np.repeat(np.arange(n), np.diff(ip))
    assert not np.any(rows == ix)
    key = rows.astype(np.int64) << 32 | ix
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.sort(ix.astype(np.int64) << 32 | rows), key)
    ip2, ix2 = oracle.rmat_csr(10, 16, seed=42, self_loops=True)
    assert ip2[-1] == ip[-1] + n
    # the id scramble is a bijection for every width
    for bits in (1, 2, 7, 12):
        x = oracle.scramble_id(np.arange(1 << bits), bits, oracle.stream_key(42, 1))
        assert np.array_equal(np.sort(x), np.arange(1 << bits))
    # skew survives the scramble
    deg = np.diff(ip)
    assert deg.max() > 20 * max(1, np.median(deg))


def test_model_oracle_gradients_match_finite_differences(oracle):
    """The analytic backward of the 2-layer GCN restatement (models.py:43-99) against central
    differences of its own float64-accurate loss, dropout off."""
    fx = load_golden('cora')
    n = int(fx['n'])
    ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    feat = (fx['feat_indptr'].astype(np.int64), fx['feat_coords'][:, 1], fx['feat_values'].astype(np.float32))
    rng = np.random.RandomState(0)
    f_in, c = int(fx['feat_shape'][1]), fx['labels'].shape[1]
    w = [oracle.glorot((f_in, 16), rng), oracle.glorot((16, c), rng)]
    model = oracle.GcnOracle((ip, ix, v32), feat, n, w)
    labels, mask = fx['y_train'], fx['train_mask']
    logits = model.forward()
    grads = model.backward(logits, labels, mask)
    base = model.loss(logits, labels, mask)
    assert np.isfinite(base)
    for li, idx in [(0, (int(feat[1][0]), 3)), (1, (5, 2)), (1, (0, 0))]:
        eps = 1e-2
        old = model.w[li][idx]
        model.w[li][idx] = old + eps
        lp = model.loss(model.forward(), labels, mask)
        model.w[li][idx] = old - eps
        lm = model.loss(model.forward(), labels, mask)
        model.w[li][idx] = old
        fd = (float(lp) - float(lm)) / (2 * eps)
        assert abs(fd - grads[li][idx]) <= 2e-2 * max(abs(fd), 1e-3) + 2e-4, (li, idx, fd, grads[li][idx])
