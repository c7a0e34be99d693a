"""The kernel variants a LARGE matrix selects, against the oracle.

A matrix whose work axis (entries + 4 per row) fills 8192 strips of 256 units takes other
template instances of the row kernel than Cora / CiteSeer / PubMed do: 256-unit strips, 4-entry
batches, one B-row batch in flight per row, rows cut every 256 (narrow operand) or 1024 (wide
operand) entries.  Those are the instances the R-MAT benchmark sizes run (gcn/layers.py:202 at
2^22 nodes x 512 columns and 2^24 x 256), so each of them is compared here with the serial loop of
the oracle on a scale-17 R-MAT A_hat: bit for bit on every row up to the split threshold, to
1e-5 * max|ref64| on the chunked hub rows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
SCALE = 17


@pytest.fixture(scope='module')
def big_hat():
    """A_hat of the scale-17 R-MAT graph on the device, its arrays on the host."""
    import torch
    from gcn_amd import sparse
    n = 1 << SCALE
    ip, ix = sparse.rmat_csr(SCALE, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(ip, None), 'sym')
    v = sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, ip, ix, v)
    # the large-matrix regime: full-length strips
    assert a.plan.n_strips >= 8192
    assert (a.nnz + 4 * n) // a.plan.n_strips >= 255
    assert a.plan.n_long_rows > 0 and a.plan.split_nnz == 256
    yield n, a, ip.cpu().numpy(), ix.cpu().numpy(), v.cpu().numpy()
    del a
    torch.cuda.empty_cache()


def _check(out, ref32, ref64, lens, split):
    short = lens <= split
    assert short.sum() > 0 and (~short).sum() > 0
    assert np.array_equal(out[short], ref32[short])
    scale = np.abs(ref64).max()
    assert np.abs(out - ref64).max() <= REL_TOL * scale
    np.testing.assert_allclose(out, ref64, rtol=REL_TOL, atol=REL_TOL * scale)


# width -> (lanes per row, floats per vector, vectors per lane) of the instance it selects
@pytest.mark.parametrize('f', [4, 8, 16, 32, 64, 128, 256, 512, 1028, 7, 33, 130])
def test_every_large_matrix_instance_against_the_serial_loop(big_hat, oracle, f):
    import torch
    from gcn_amd import sparse
    n, a, ip, ix, v = big_hat
    b = sparse.synthetic_dense(n, f, seed=13)
    bn = b.cpu().numpy()
    assert np.array_equal(bn, oracle.synthetic_dense(n, f, seed=13))
    ref32 = oracle.spmm_csr(ip, ix, v, bn, threads=8)
    ref64 = oracle.spmm_csr_f64(ip, ix, v, bn)
    out = sparse.spmm(a, b)
    vectorised = f % 4 == 0
    wide = (f // 4 if vectorised else f) > 16          # 32 or 64 lanes per row: rows cut at 1024 entries
    split = 1024 if wide else 256
    _check(out.cpu().numpy(), ref32, ref64, np.diff(ip), split)
    # ReLU epilogue, strip rows and the long-row reduction alike
    assert torch.equal(sparse.spmm(a, b, relu=True), torch.clamp(out, min=0))
    # a second call gives the same bits (chunk sums are added in a fixed order)
    assert torch.equal(sparse.spmm(a, b), out)


@pytest.mark.parametrize('f', [16, 128, 256, 512])
def test_addend_and_store_dropout_forms_on_the_large_matrix(big_hat, oracle, f):
    import torch
    from gcn_amd import sparse
    n, a, ip, ix, v = big_hat
    b = sparse.synthetic_dense(n, f, seed=14)
    plain = sparse.spmm(a, b)
    # C = A.B + D (gcn/utils.py:568-569): the product's bits plus one more addition
    dmat = sparse.synthetic_dense(n, f, seed=15)
    assert torch.equal(sparse.spmm_add(a, b, dmat), plain + dmat)
    # dropout(relu(A.B)) in the store = the oracle's dense dropout of relu(A.B)
    keep_prob = 0.5
    fused = sparse.spmm(a, b, relu=True, out_dropout=sparse.OutDropout(keep_prob, 31, 2))
    ref, _ = oracle.dense_dropout(np.maximum(plain.cpu().numpy(), 0), keep_prob, oracle.stream_key(31, 2))
    assert np.array_equal(fused.cpu().numpy(), ref)
    # a column range with the long rows cut as the full width cuts them
    if f >= 128:
        c0, c1 = 32, 96
        part = sparse.spmm(a, b[:, c0:c1], wide_split=True)
        assert torch.equal(part, plain[:, c0:c1])
    # the addend may be the output itself, whole or a column range; activation and store dropout after the sum
    summed = plain + dmat
    inplace = dmat.clone()
    assert sparse.spmm_add(a, b, inplace, out=inplace) is inplace and torch.equal(inplace, summed)
    if f >= 128:
        inplace = dmat.clone()
        for c0, c1 in ((0, 32), (32, 96), (96, f)):
            sparse.spmm_add(a, b[:, c0:c1], inplace[:, c0:c1], out=inplace[:, c0:c1], wide_split=True)
        assert torch.equal(inplace, summed)
    inplace = dmat.clone()
    sparse.spmm_add(a, b, inplace, relu=True, out=inplace, out_dropout=sparse.OutDropout(keep_prob, 31, 2, index_offset=8 * f))
    ref, _ = oracle.dense_dropout(np.maximum(summed.cpu().numpy(), 0), keep_prob, oracle.stream_key(31, 2),
                                  index_offset=8 * f)
    assert np.array_equal(inplace.cpu().numpy(), ref)


@pytest.mark.parametrize('f', [16, 64, 128, 256, 512])
def test_sparse_dropout_form_on_the_large_matrix(big_hat, oracle, f):
    """sparse_dropout (gcn/layers.py:19-25) fused on the entries of a large matrix: the instances with
    DROPOUT = true and 4-entry batches."""
    from gcn_amd import sparse
    n, a, ip, ix, v = big_hat
    keep_prob, seed, stream_id = 0.5, 77, 3
    keep, scaled = oracle.sparse_dropout_values(v, keep_prob, oracle.stream_key(seed, stream_id))
    rows = np.repeat(np.arange(n), np.diff(ip))
    ip2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ip2, rows[keep] + 1, 1)
    ip2 = np.cumsum(ip2)
    b = sparse.synthetic_dense(n, f, seed=16)
    bn = b.cpu().numpy()
    ref32 = oracle.spmm_csr(ip2, ix[keep], scaled[keep], bn, threads=8)
    ref64 = oracle.spmm_csr_f64(ip2, ix[keep], scaled[keep], bn)
    out = sparse.spmm_dropout(a, b, keep_prob, seed, stream_id).cpu().numpy()
    # the kernel cuts rows by their stored length, kept or not
    split = 1024 if f // 4 > 16 else 256
    _check(out, ref32, ref64, np.diff(ip), split)


def test_scale_22_width_512_properties(oracle):
    """The 2^22-node, 512-column roofline-sweep size (R-MAT, average degree 32) through properties that need no full CPU
    product: exact linearity, sampled rows against the serial loop, operator symmetry, A.1."""
    import torch
    from gcn_amd import sparse
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip('needs 60 GiB of free HBM')
    scale, f = 22, 512
    n = 1 << scale
    ip, ix = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(ip, None), 'sym')
    v = sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, ip, ix, v)
    b = sparse.synthetic_dense(n, f, seed=7)
    c = sparse.spmm(a, b)
    c2 = sparse.spmm(a, b * 2)
    assert torch.equal(c2, c * 2)
    del c2
    rng = np.random.default_rng(6)
    lens = ip[1:] - ip[:-1]
    longest = torch.topk(lens, 4).indices.cpu().numpy()
    sample = np.unique(np.concatenate([rng.integers(0, n, 3000), longest, [0, n - 1]]))
    ipn = ip.cpu().numpy()
    starts, ends = ipn[sample], ipn[sample + 1]
    take = torch.from_numpy(np.concatenate([np.arange(s, e) for s, e in zip(starts, ends)])).cuda()
    uniq, inv = torch.unique(ix[take].to(torch.int64), return_inverse=True)
    mini_ip = np.concatenate([[0], np.cumsum(ends - starts)]).astype(np.int64)
    mini_ix = inv.cpu().numpy().astype(np.int32)
    mini_v = v[take].cpu().numpy()
    mini_b = b[uniq].cpu().numpy()
    ref32 = oracle.spmm_csr(mini_ip, mini_ix, mini_v, mini_b, threads=8)
    ref64 = oracle.spmm_csr_f64(mini_ip, mini_ix, mini_v, mini_b)
    got = c[torch.from_numpy(sample).cuda()].cpu().numpy()
    short = (ends - starts) <= 1024
    assert short.sum() >= 2900 and (~short).sum() >= 4
    assert np.array_equal(got[short], ref32[short])
    assert np.abs(got - ref64).max() <= REL_TOL * np.abs(ref64).max()
    y = sparse.synthetic_dense(n, f, seed=11)
    cy = sparse.spmm(a, y)
    lhs = float((y.double() * c.double()).sum())
    rhs = float((cy.double() * b.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    del y, cy, c, b
    ones = torch.ones((n, 8), dtype=torch.float32, device='cuda')
    s = sparse.spmm(a, ones)
    assert bool((s == s[:, :1]).all())
    row_sum = sparse.row_sums(ip, v.double())
    assert float((s[:, 0].double() - row_sum).abs().max()) <= REL_TOL * float(row_sum.abs().max())
    del a
    torch.cuda.empty_cache()


@pytest.mark.parametrize('f', [4, 8, 12, 16, 32, 64])
def test_lds_staged_hub_rows_change_no_bit(big_hat, oracle, f):
    """The LDS-staged form of a narrow product (gcn_spmm_plan_set_lds_hub_rows) on the degree-ordered
    matrix: hub rows of B served from LDS, the same additions in the same order -- identical bits to the
    plain form, which is itself checked against the oracle on the same matrix."""
    import torch
    from gcn_amd import sparse
    n, a, ip, ix, v = big_hat
    new_id, old_id = sparse.degree_order(a.indptr)
    r_ip, r_ix, r_v = sparse.csr_relabel(n, n, a.indptr, a.indices, a.values.double(), new_id, new_id)
    ordered = sparse.DeviceCSR(n, n, r_ip, r_ix.contiguous(), r_v.to(torch.float32))
    b = sparse.synthetic_dense(n, f, seed=21)
    plain = sparse.spmm(ordered, b)
    ref32 = oracle.spmm_csr(r_ip.cpu().numpy(), r_ix.cpu().numpy(), ordered.values.cpu().numpy(), b.cpu().numpy(),
                            threads=8)
    short = np.diff(r_ip.cpu().numpy()) <= 256
    assert np.array_equal(plain.cpu().numpy()[short], ref32[short])
    for hubs in (1, 100, 512, 1 << 20):                    # more than fit: clipped to 32 KiB of rows
        ordered.plan.set_lds_hub_rows(hubs)
        assert torch.equal(sparse.spmm(ordered, b), plain)
        assert torch.equal(sparse.spmm(ordered, b, relu=True), torch.clamp(plain, min=0))
    ordered.plan.set_lds_hub_rows(0)
    # a strided operand (row pitch wider than f) stages the same rows
    wide = torch.zeros((n, f + 4), device='cuda')
    wide[:, :f] = b
    ordered.plan.set_lds_hub_rows(256)
    assert torch.equal(sparse.spmm(ordered, wide[:, :f]), plain)
    ordered.plan.set_lds_hub_rows(0)


def test_lds_hub_rows_on_an_empty_matrix():
    """A large matrix without entries reads no row of B -- the operand may even be absent -- so a plan with
    LDS hub rows switched on must not stage any: the product is all zeros."""
    import ctypes
    import torch
    from gcn_amd import _lib, sparse
    n, f = 1 << 20, 16
    empty = sparse.DeviceCSR(n, n, torch.zeros(n + 1, dtype=torch.int64, device='cuda'),
                             torch.zeros(0, dtype=torch.int32, device='cuda'), torch.zeros(0, device='cuda'))
    empty.plan.set_lds_hub_rows(512)
    b = sparse.synthetic_dense(n, f, seed=2)
    out = torch.full((n, f), float('nan'), device='cuda')
    sparse.spmm(empty, b, out=out)
    assert bool((out == 0).all())
    out.fill_(float('nan'))
    # the C ABI accepts a null operand for a matrix without entries
    _lib.call('gcn_spmm_csr_f32', empty.plan.handle, ctypes.c_void_p(empty.indptr.data_ptr()), None, None, f, None, f,
              ctypes.c_void_p(out.data_ptr()), f, 0, None, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bool((out == 0).all())


@pytest.mark.parametrize('f', [4, 16, 32, 64, 128, 256, 512, 640])
def test_operand_in_two_parts_gives_the_bits_of_the_stacked_operand(big_hat, f):
    """gcn_spmm_csr_two_part_f32: the dense operand as [first; rest], read in place -- what a rank of the
    row-partitioned product does with its own rows and the rows it received.  Every width class of the
    large-matrix kernels, with the ReLU and the store dropout, parts of different row pitch, an empty second
    part, and the layouts the two-part form does not take (which the wrapper stacks)."""
    import torch
    from gcn_amd import sparse
    n, a, ip, ix, v = big_hat
    b = sparse.synthetic_dense(n, f, seed=41)
    whole = sparse.spmm(a, b)
    whole_relu = sparse.spmm(a, b, relu=True)
    for n_first in (1, n // 3, n - 5, n):
        first, rest = b[:n_first].clone(), b[n_first:].clone()
        assert torch.equal(sparse.spmm(a, rest, first=first), whole)
        assert torch.equal(sparse.spmm(a, rest if n_first < n else None, first=first, relu=True), whole_relu)
    n_first = n // 3
    # parts that are column ranges of wider matrices (row pitch > f), output a column range too
    wide_first = torch.zeros((n_first, f + 8), device='cuda')
    wide_first[:, 4:4 + f] = b[:n_first]
    wide_rest = torch.zeros((n - n_first, f + 12), device='cuda')
    wide_rest[:, 8:8 + f] = b[n_first:]
    out = torch.full((n, f + 4), -2.0, device='cuda')
    sparse.spmm(a, wide_rest[:, 8:8 + f], first=wide_first[:, 4:4 + f], out=out[:, 4:])
    assert torch.equal(out[:, 4:], whole) and bool((out[:, :4] == -2.0).all())
    drop = sparse.OutDropout(0.5, 17, 3, index_offset=8 * f, pitch=f)
    assert torch.equal(sparse.spmm(a, b[n_first:], first=b[:n_first], relu=True, out_dropout=drop),
                       sparse.spmm(a, b, relu=True, out_dropout=drop))
    assert torch.equal(sparse.spmm(a, b[n_first:], first=b[:n_first], wide_split=True), sparse.spmm(a, b, wide_split=True))
    # a layout without 128-bit rows: the wrapper stacks the parts
    odd = torch.zeros((n_first, f + 1), device='cuda')
    odd[:, 1:] = b[:n_first]
    assert torch.equal(sparse.spmm(a, b[n_first:], first=odd[:, 1:]), whole)
    with pytest.raises(ValueError, match='shape mismatch'):
        sparse.spmm(a, b[n_first + 1:], first=b[:n_first])
