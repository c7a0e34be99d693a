"""The headline size -- R-MAT scale 24, edge factor 16, F = 256 -- checked through properties that
do not need a full-size CPU product: structure invariants of the generated CSR, Â·1 against the
independent row-sum kernel, exact linearity, the symmetry identity <y, Âx> = <Ây, x>, and a sample
of rows compared bit for bit with the oracle's serial loop on the rows of B they reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCALE, EDGE_FACTOR, F = 24, 16, 256


@pytest.fixture(scope='module')
def big():
    import torch
    from gcn_amd import sparse
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip('needs 120 GiB of free HBM')
    n = 1 << SCALE
    indptr, indices = sparse.rmat_csr(SCALE, EDGE_FACTOR, seed=42, self_loops=True)
    deg = sparse.row_sums(indptr, None)
    d = sparse.degree_scale(deg, 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    yield n, a, deg, d
    del a
    torch.cuda.empty_cache()


def test_structure_invariants(big):
    import torch
    n, a, deg, d = big
    ip, ix = a.indptr, a.indices
    assert int(ip[0]) == 0 and int(ip[-1]) == a.nnz and bool((ip[1:] > ip[:-1]).all())      # +I: no empty row
    rows = torch.repeat_interleave(torch.arange(n, device='cuda', dtype=torch.int32), ip[1:] - ip[:-1])
    # strictly ascending columns inside every row: sorted and free of duplicates
    same_row = rows[1:] == rows[:-1]
    assert bool((ix[1:][same_row] > ix[:-1][same_row]).all())
    assert int((rows == ix).sum()) == n                                                    # one diagonal entry per row
    assert bool((ix >= 0).all()) and bool((ix < n).all())
    # simple undirected graph: as many entries (r, c) as (c, r) -- column counts equal row counts
    col_counts = torch.bincount(ix.to(torch.int64), minlength=n)
    assert torch.equal(col_counts, ip[1:] - ip[:-1])
    assert torch.equal(deg, (ip[1:] - ip[:-1]).to(torch.float64))
    del rows, same_row, col_counts
    # the transpose (stable sort by column) reproduces the matrix: pattern and float32 values symmetric
    at = a.transpose()
    assert torch.equal(at.indptr, a.indptr) and torch.equal(at.indices, a.indices)
    assert torch.equal(at.values, a.values)                                                # float32 values exactly symmetric
    a._t = None                                                                            # release the cached copy


def test_product_properties_at_full_size(big, oracle):
    import torch
    from gcn_amd import sparse
    n, a, deg, d = big
    b = sparse.synthetic_dense(n, F, seed=7)
    c = sparse.spmm(a, b)
    # 1. exact linearity (scaling by a power of two commutes with every rounding)
    c2 = sparse.spmm(a, b * 2)
    assert torch.equal(c2, c * 2)
    del c2
    # 2. sampled rows against the serial loop, on the rows of B they reference
    rng = np.random.default_rng(5)
    lens = (a.indptr[1:] - a.indptr[:-1])
    longest = torch.topk(lens, 4).indices.cpu().numpy()
    sample = np.unique(np.concatenate([rng.integers(0, n, 2000), longest, [0, n - 1]]))
    ipn = a.indptr.cpu().numpy()
    starts, ends = ipn[sample], ipn[sample + 1]
    take = np.concatenate([np.arange(s, e) for s, e in zip(starts, ends)])
    take_t = torch.from_numpy(take).cuda()
    cols = a.indices[take_t].to(torch.int64)
    uniq, inv = torch.unique(cols, return_inverse=True)
    mini_ip = np.concatenate([[0], np.cumsum(ends - starts)]).astype(np.int64)
    ref32 = oracle.spmm_csr(mini_ip, inv.cpu().numpy().astype(np.int32), a.values[take_t].cpu().numpy(),
                            b[uniq].cpu().numpy())
    ref64 = oracle.spmm_csr_f64(mini_ip, inv.cpu().numpy().astype(np.int32), a.values[take_t].cpu().numpy(),
                                b[uniq].cpu().numpy())
    got = c[torch.from_numpy(sample).cuda()].cpu().numpy()
    short = (ends - starts) <= 1024            # one wave per row at this width: rows up to 1024 entries are serial
    assert short.sum() >= 1900 and (~short).sum() >= 4
    assert np.array_equal(got[short], ref32[short])
    assert np.abs(got - ref64).max() <= 1e-5 * np.abs(ref64).max()
    # 3. symmetry of the operator: <y, A x> == <A y, x> (float64 accumulation of the dot products)
    y = sparse.synthetic_dense(n, F, seed=11)
    cy = sparse.spmm(a, y)
    lhs = float((y.double() * c.double()).sum())
    rhs = float((cy.double() * b.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    del y, cy
    # 4. A . 1 = row sums of the values, every column the same bits
    ones = torch.ones((n, 8), dtype=torch.float32, device='cuda')
    s = sparse.spmm(a, ones)
    assert bool((s == s[:, :1]).all())
    row_sum = sparse.row_sums(a.indptr, a.values.double())
    assert float((s[:, 0].double() - row_sum).abs().max()) <= 1e-5 * float(row_sum.abs().max())


def test_packed_operand_products_at_full_size(big):
    """The products a training step runs on operands that are mostly zeros, at the headline size: A_hat times
    dropout(X) gathered in packed form equals, bit for bit, the dense kernel's product with the dropped matrix (its
    long rows cut where the wide product cuts them) at half and at a quarter kept; the gated gradient formed for a
    list of rows only equals the one formed for every row."""
    import torch
    from gcn_amd import sparse
    n, a, deg, d = big
    x = sparse.synthetic_dense(n, F, seed=7)
    out_dense = torch.empty((n, F), device='cuda')
    out_packed = torch.empty((n, F), device='cuda')
    packed = sparse.PackedRows.empty(n, F, x.device)
    for keep in (0.5, 0.25):
        dropped = sparse.dropout(x, keep, 5, 1)
        sparse.dropout_pack(x, keep, 5, 1, out=packed)
        assert int(packed.kept().sum()) == int((dropped != 0).sum())
        sparse.spmm(a, dropped, out=out_dense, wide_split=True)
        sparse.spmm_packed(a, packed, out=out_packed)
        assert torch.equal(out_dense, out_packed)
        del dropped
    # the hidden layer's gradient, non-zero in one row of eight
    live = torch.arange(0, n, 8, device='cuda')
    dp = torch.zeros((n, 16), device='cuda')
    dp[live] = sparse.synthetic_dense(live.numel(), 16, seed=8)
    w = sparse.synthetic_dense(F, 16, seed=9)
    gate = torch.clamp(x, min=0)
    del x
    every = sparse.gemm_gated_pack(dp, w, gate, 0.5)
    listed = sparse.gemm_gated_pack(dp, w, gate, 0.5, out=packed, rows=live)
    assert torch.equal(every.kept(), listed.kept())
    sparse.spmm_packed(a, every, out=out_dense)
    sparse.spmm_packed(a, listed, out=out_packed)
    assert torch.equal(out_dense, out_packed)
    assert bool((listed.kept()[live] > 0).all())
