"""The packed dense operand (rows stored without their zeros) and the product that gathers it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sparse_dense(n, f, density, seed):
    import torch
    from gcn_amd import sparse
    x = sparse.synthetic_dense(n, f, seed=seed)
    pick = sparse.synthetic_dense(n, f, seed=seed + 100)
    lo, hi = float(pick.min()), float(pick.max())
    return torch.where(pick < lo + density * (hi - lo), x, torch.zeros_like(x))


@pytest.mark.parametrize('f', [256, 192, 130, 33, 7])
@pytest.mark.parametrize('density', [0.0, 0.25, 0.5, 1.0])
def test_pack_and_unpack_round_trip(f, density):
    import torch
    from gcn_amd import sparse
    n = 3001
    x = _sparse_dense(n, f, density, seed=5)
    if density == 1.0:
        x = torch.where(x == 0, torch.ones_like(x), x)
    wide = torch.zeros((n, f + 3), device='cuda')             # a strided, unaligned source
    wide[:, 1:1 + f] = x
    for src in (x, wide[:, 1:1 + f]):
        packed = sparse.pack_rows(src)
        assert packed.ld % 32 == 0 and packed.ld >= 16 + f
        assert torch.equal(packed.dense(), x)
        assert torch.equal(packed.kept(), (x != 0).sum(dim=1))
    # the header of a row: eight {mask, kept columns in the words before} pairs
    head = packed.data[:, :16].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    bits = (x != 0).cpu().numpy()
    for w in range(8):
        cols = bits[:, 32 * w:32 * w + 32]
        want = (cols * (1 << np.arange(cols.shape[1], dtype=np.int64))).sum(axis=1)
        assert np.array_equal(head[:, 2 * w], want)
        assert np.array_equal(head[:, 2 * w + 1], bits[:, :32 * w].sum(axis=1))


@pytest.mark.parametrize('keep_prob', [0.5, 0.9, 0.1])
@pytest.mark.parametrize('f', [256, 100])
def test_dropout_straight_into_the_packed_form(oracle, f, keep_prob):
    import torch
    from gcn_amd import sparse
    n, offset = 2050, 12 * f
    x = sparse.synthetic_dense(n, f, seed=8)
    want = sparse.dropout(x, keep_prob, 77, 3, index_offset=offset)
    ref, _ = oracle.dense_dropout(x.cpu().numpy(), keep_prob, oracle.stream_key(77, 3), index_offset=offset)
    assert np.array_equal(want.cpu().numpy(), ref)
    packed = sparse.dropout_pack(x, keep_prob, 77, 3, index_offset=offset)
    assert torch.equal(packed.dense(), want)


@pytest.mark.parametrize('f', [256, 160, 64])
@pytest.mark.parametrize('density', [0.5, 0.25, 0.02])
def test_product_with_a_packed_operand_gives_the_dense_product(oracle, f, density):
    """Scale-17 R-MAT A_hat (rows of every length, chunked hub rows included) times a matrix that is mostly zeros:
    the packed product against the dense kernel on the expanded matrix -- equal as numbers everywhere (the dense
    kernel cuts its long rows at the same entries when told to split as the wide product does) -- and against the
    CPU loop on the rows it sums serially."""
    import torch
    from gcn_amd import sparse
    scale = 17
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    b = _sparse_dense(n, f, density, seed=21)
    packed = sparse.pack_rows(b)
    got = sparse.spmm_packed(a, packed)
    dense = sparse.spmm(a, b, wide_split=True)
    assert torch.equal(got, dense)
    assert torch.equal(sparse.spmm_packed(a, packed, relu=True), torch.clamp(dense, min=0))
    ip, ix, v = indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy()
    cpu = oracle.spmm_csr(ip, ix, v, b.cpu().numpy(), threads=8)
    short = np.diff(ip) <= 1024
    assert short.sum() < n and np.array_equal(got.cpu().numpy()[short], cpu[short])
    # a strided output
    wide = torch.full((n, f + 8), -2.0, device='cuda')
    sparse.spmm_packed(a, packed, out=wide[:, 4:4 + f])
    assert torch.equal(wide[:, 4:4 + f], dense) and bool((wide[:, :4] == -2.0).all()) and bool((wide[:, 4 + f:] == -2.0).all())


def test_small_matrix_and_empty_rows_with_a_packed_operand(oracle):
    import torch
    from gcn_amd import sparse
    from conftest import load_golden
    fx = load_golden('cora')
    n = int(fx['n'])
    a = sparse.DeviceCSR(n, n, torch.as_tensor(fx['ai_indptr'].astype(np.int64)).cuda(),
                         torch.as_tensor(fx['ai_indices'].astype(np.int32)).cuda(),
                         torch.as_tensor(fx['hat_values'].astype(np.float32)).cuda())
    b = _sparse_dense(n, 256, 0.3, seed=4)
    assert torch.equal(sparse.spmm_packed(a, sparse.pack_rows(b)), sparse.spmm(a, b, wide_split=True))
    empty = sparse.DeviceCSR(5, n, torch.zeros(6, dtype=torch.int64, device='cuda'),
                             torch.zeros(0, dtype=torch.int32, device='cuda'), torch.zeros(0, device='cuda'))
    assert bool((sparse.spmm_packed(empty, sparse.pack_rows(b)) == 0).all())
    with pytest.raises(ValueError):
        sparse.pack_rows(torch.zeros((4, 300), device='cuda'))


@pytest.mark.parametrize('m,n,k', [(5000, 256, 16), (3001, 256, 7), (2050, 160, 16), (700, 33, 1)])
def test_gated_gradient_straight_into_packed_rows(oracle, m, n, k):
    """where(gate > 0, (A . B^T) / keep, 0) written in packed form: the zero pattern of the gate exactly, the values
    of gcn_gemm_gated_f32 up to the order of the k additions, and of a float64 evaluation to 1e-6."""
    import torch
    from gcn_amd import sparse
    a = torch.as_tensor(oracle.synthetic_dense(m, k, seed=81)).cuda()
    b = torch.as_tensor(oracle.synthetic_dense(n, k, seed=82)).cuda()
    gate = _sparse_dense(m, n, 0.5, seed=83)
    gate = torch.where(gate > 0, gate, torch.zeros_like(gate))          # zeros and positives, as dropped activations
    for keep in (0.5, 0.8):
        packed = sparse.gemm_gated_pack(a, b, gate, keep)
        got = packed.dense()
        tiled = sparse.gemm_gated(a, b, gate, keep, trans_b=True)
        exact = torch.where(gate > 0, (a.double() @ b.double().t()) / keep, torch.zeros((m, n), dtype=torch.float64, device='cuda'))
        scale = float(exact.abs().max())
        assert bool((got[gate <= 0] == 0).all())
        assert float((got.double() - exact).abs().max()) <= 1e-6 * scale
        assert float((got - tiled).abs().max()) <= 1e-6 * scale
        assert torch.equal(packed.kept(), (got != 0).sum(dim=1))
    # strided operands
    a_wide = torch.zeros((m, k + 3), device='cuda')
    a_wide[:, 1:1 + k] = a
    g_wide = torch.zeros((m, n + 5), device='cuda')
    g_wide[:, 2:2 + n] = gate
    assert torch.equal(sparse.gemm_gated_pack(a_wide[:, 1:1 + k], b, g_wide[:, 2:2 + n], 0.8).dense(), got)
    with pytest.raises(ValueError):
        sparse.gemm_gated_pack(torch.zeros((4, 17), device='cuda'), torch.zeros((8, 17), device='cuda'),
                               torch.zeros((4, 8), device='cuda'), 0.5)


@pytest.mark.parametrize('f', [255, 130, 33, 4, 1])
def test_packed_product_at_ragged_widths(oracle, f):
    """Widths that are not multiples of 4 or of a mask word, on a scale-12 graph with a strided output."""
    import torch
    from gcn_amd import sparse
    scale = 12
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=3, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    b = _sparse_dense(n, f, 0.4, seed=9)
    ref = oracle.spmm_csr(indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy(), b.cpu().numpy())
    packed = sparse.pack_rows(b)
    got = sparse.spmm_packed(a, packed)
    short = np.diff(indptr.cpu().numpy()) <= 1024
    assert np.array_equal(got.cpu().numpy()[short], ref[short])
    assert float(np.abs(got.cpu().numpy() - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    wide = torch.full((n, f + 3), 7.0, device='cuda')
    sparse.spmm_packed(a, packed, out=wide[:, 1:1 + f])
    assert torch.equal(wide[:, 1:1 + f], got) and bool((wide[:, 0] == 7.0).all()) and bool((wide[:, 1 + f:] == 7.0).all())


@pytest.mark.parametrize('f', [256, 100])
def test_packed_product_over_hand_made_row_lengths(oracle, f):
    """The strip part of the packed product walks consecutive rows as one stream of entries: row ends inside a
    batch, at a batch boundary, rows without entries (first, last, several in a row), and rows beyond the split
    threshold (summed chunk-wise) in front of, between and behind short ones.  Every short row bit for bit against
    the CPU loop, the whole result against the dense kernel."""
    import torch
    from gcn_amd import sparse
    rng = np.random.RandomState(17)
    n_cols = 5000
    pattern = [0, 0, 1, 4, 8, 3, 0, 5, 12, 1500, 2, 0, 0, 0, 7, 1024, 1025, 1, 4, 4, 4, 2100, 3000, 9, 0, 16, 31, 33,
               0, 1, 1, 1, 2, 600, 255, 257, 0]
    lengths = np.array(pattern * 40 + [0, 0], dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, size=int(k), replace=False)) for k in lengths]).astype(np.int32)
    vals = rng.uniform(-1, 1, size=indices.size).astype(np.float32)
    a = sparse.DeviceCSR(lengths.size, n_cols, torch.as_tensor(indptr).cuda(), torch.as_tensor(indices).cuda(),
                         torch.as_tensor(vals).cuda())
    b = _sparse_dense(n_cols, f, 0.3, seed=5)
    got = sparse.spmm_packed(a, sparse.pack_rows(b))
    assert torch.equal(got, sparse.spmm(a, b, wide_split=True))
    assert torch.equal(sparse.spmm_packed(a, sparse.pack_rows(b), relu=True),
                       torch.clamp(sparse.spmm(a, b, wide_split=True), min=0))
    cpu = oracle.spmm_csr(indptr, indices, vals, b.cpu().numpy())
    short = lengths <= 1024
    assert np.array_equal(got.cpu().numpy()[short], cpu[short])
    assert bool((got[torch.as_tensor(lengths == 0).cuda()] == 0).all())


@pytest.mark.parametrize('f', [256, 160])
def test_dropout_into_packed_rows_by_row_number(oracle, f):
    """Rows held out of place draw the decisions of the rows they are: packing a subset of the rows of X with their
    row numbers gives those rows of dropout(X), whatever order they are held in."""
    import torch
    from gcn_amd import sparse
    n = 3000
    x = torch.as_tensor(oracle.synthetic_dense(n, f, seed=31)).cuda()
    whole = sparse.dropout(x, 0.5, 7, 3)
    assert torch.equal(sparse.dropout_pack(x, 0.5, 7, 3).dense(), whole)
    gen = torch.Generator().manual_seed(2)
    rows = torch.randperm(n, generator=gen)[:1111].cuda()
    held = x[rows].contiguous()
    assert torch.equal(sparse.dropout_pack(held, 0.5, 7, 3, row_ids=rows).dense(), whole[rows])
    # both parts of one buffer, as a rank packs [own rows | rows of other ranks]
    buf = sparse.PackedRows.empty(n + rows.numel(), f, x.device)
    sparse.dropout_pack(x, 0.5, 7, 3, out=buf.rows(0, n))
    sparse.dropout_pack(held, 0.5, 7, 3, row_ids=rows, out=buf.rows(n, n + rows.numel()))
    assert torch.equal(buf.dense(), torch.cat([whole, whole[rows]]))
    with pytest.raises(ValueError):
        sparse.dropout_pack(held, 0.5, 7, 3, row_ids=rows[:5])


def test_gated_gradient_for_listed_rows_only(oracle):
    """The gradient is zero outside some rows: with those rows listed, the packed result equals the one formed from
    every row (the unlisted rows come out empty), whatever the buffer held before; out-of-range row numbers are
    skipped; an empty list empties every row."""
    import torch
    from gcn_amd import sparse
    m, n, k = 6000, 256, 16
    gen = torch.Generator().manual_seed(3)
    rows = torch.sort(torch.randperm(m, generator=gen)[:700])[0].cuda()
    a = torch.zeros((m, k), device='cuda')
    a[rows] = torch.as_tensor(oracle.synthetic_dense(rows.numel(), k, seed=85)).cuda()
    b = torch.as_tensor(oracle.synthetic_dense(n, k, seed=86)).cuda()
    gate = torch.clamp(_sparse_dense(m, n, 0.5, seed=87), min=0)
    whole = sparse.gemm_gated_pack(a, b, gate, 0.5)
    stale = sparse.PackedRows.empty(m, n, a.device)
    stale.data.fill_(float('nan'))
    listed = sparse.gemm_gated_pack(a, b, gate, 0.5, out=stale, rows=rows)
    assert torch.equal(listed.dense(), whole.dense()) and torch.equal(listed.kept(), whole.kept())
    assert bool((listed.kept()[rows] > 0).any())
    # a row number that is not a row of the matrix is skipped
    padded = torch.cat([torch.tensor([-1], device='cuda'), rows, torch.tensor([m, m + 5], device='cuda')])
    with pytest.raises(ValueError):                       # more rows listed than the matrix has
        sparse.gemm_gated_pack(a, b, gate, 0.5, out=stale, rows=torch.arange(m + 1, device='cuda'))
    assert torch.equal(sparse.gemm_gated_pack(a, b, gate, 0.5, out=stale, rows=padded).dense(), whole.dense())
    none = sparse.gemm_gated_pack(a, b, gate, 0.5, out=stale, rows=rows[:0])
    assert bool((none.dense() == 0).all()) and int(none.kept().sum()) == 0


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_packed_product_on_random_row_length_mixes(oracle, seed):
    """Random matrices with heavy-tailed row lengths (many empty and one-entry rows, some beyond the split threshold),
    random widths and densities: the packed product equals the dense kernel's, bit for bit, and the CPU loop on every
    row it sums serially."""
    import torch
    from gcn_amd import sparse
    rng = np.random.RandomState(100 + seed)
    n_rows, n_cols = int(rng.randint(2000, 9000)), int(rng.randint(3000, 12000))
    lengths = np.minimum((rng.pareto(0.9, size=n_rows) * 2).astype(np.int64), n_cols)
    lengths[rng.rand(n_rows) < 0.2] = 0
    lengths[rng.randint(0, n_rows, size=3)] = rng.randint(1025, min(n_cols, 3000), size=3)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, size=int(k), replace=False)) if k else np.zeros(0, np.int64)
                              for k in lengths]).astype(np.int32)
    vals = rng.uniform(-1, 1, size=indices.size).astype(np.float32)
    a = sparse.DeviceCSR(n_rows, n_cols, torch.as_tensor(indptr).cuda(), torch.as_tensor(indices).cuda(),
                         torch.as_tensor(vals).cuda())
    f = int(rng.choice([256, 255, 200, 129, 64, 17]))
    b = _sparse_dense(n_cols, f, float(rng.choice([0.05, 0.25, 0.5, 0.9])), seed=seed)
    got = sparse.spmm_packed(a, sparse.pack_rows(b), relu=bool(seed & 1))
    want = sparse.spmm(a, b, wide_split=True, relu=bool(seed & 1))
    assert torch.equal(got, want)
    cpu = oracle.spmm_csr(indptr, indices, vals, b.cpu().numpy(), relu=bool(seed & 1))
    short = lengths <= 1024
    assert np.array_equal(got.cpu().numpy()[short], cpu[short])
