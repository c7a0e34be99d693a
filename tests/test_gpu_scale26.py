"""The 2^26-node graph of the row-partitioned configuration (R-MAT scale 26, edge factor 16): more than
2^31 stored entries, so row pointers beyond the int32 range and entry positions that do not fit 32 bits.

Nothing here needs a full-size CPU product: the whole graph is generated on one GPU and checked through
structure invariants, exact linearity, A_hat . 1 against the row-sum kernel, the symmetry identity and
sampled rows -- rows whose entries lie beyond position 2^31 and the four longest rows among them --
against the serial loop of the oracle on the rows of B they reference.  Then the last of eight row blocks
is generated on its own (what rank 7 of 8 builds), compared with that slice of the whole graph bit for
bit, and multiplied through the compact-operand form of the exchange plan against the matching rows of
the whole product.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCALE, EDGE_FACTOR = 26, 16
NEEDED_GIB = 110        # generation sorts 2.2 G 64-bit keys in three buffers (53 GB) beside the outputs


@pytest.fixture(scope='module')
def graph26():
    import torch
    from gcn_amd import sparse
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < NEEDED_GIB * 2 ** 30:
        pytest.skip('needs %d GiB of free HBM' % NEEDED_GIB)
    n = 1 << SCALE
    indptr, indices = sparse.rmat_csr(SCALE, EDGE_FACTOR, seed=42, self_loops=True)
    torch.cuda.empty_cache()
    deg = sparse.row_sums(indptr, None)
    d = sparse.degree_scale(deg, 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    yield n, a, deg, d
    del a, indptr, indices, vals
    torch.cuda.empty_cache()


def _dot64(x, y, rows_per_block=1 << 22):
    """<x, y> with float64 accumulation, taken over row blocks (a float64 copy of a whole operand would
    not fit beside the graph)."""
    total = 0.0
    for r in range(0, x.shape[0], rows_per_block):
        total += float((x[r:r + rows_per_block].double() * y[r:r + rows_per_block].double()).sum())
    return total


def _equals_twice(c2, c, rows_per_block=1 << 22):
    import torch
    return all(torch.equal(c2[r:r + rows_per_block], c[r:r + rows_per_block] * 2)
               for r in range(0, c.shape[0], rows_per_block))


def _sampled_rows_against_the_oracle(oracle, a, b, c, sample, serial_up_to):
    """Rows ``sample`` of c = a . b against the oracle's serial float32 loop (bit for bit on rows of at
    most ``serial_up_to`` entries) and its float64 sum (1e-5 of the largest value) on the rows of b they read."""
    import torch
    ip_s = a.indptr[torch.from_numpy(sample).cuda()].cpu().numpy()
    ip_e = a.indptr[torch.from_numpy(sample + 1).cuda()].cpu().numpy()
    lens = ip_e - ip_s
    mini_ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # entry positions of the sampled rows, built on the device (positions exceed 2^31)
    starts = torch.from_numpy(ip_s).cuda()
    lens_t = torch.from_numpy(lens).cuda()
    first = torch.from_numpy(mini_ip[:-1]).cuda()
    take = torch.repeat_interleave(starts - first, lens_t) + torch.arange(int(mini_ip[-1]), device='cuda')
    cols = a.indices[take].to(torch.int64)
    uniq, inv = torch.unique(cols, return_inverse=True)
    vals = a.values[take].cpu().numpy()
    inv = inv.cpu().numpy().astype(np.int32)
    b_rows = b[uniq].cpu().numpy()
    ref32 = oracle.spmm_csr(mini_ip, inv, vals, b_rows)
    ref64 = oracle.spmm_csr_f64(mini_ip, inv, vals, b_rows)
    got = c[torch.from_numpy(sample).cuda()].cpu().numpy()
    serial = lens <= serial_up_to
    assert np.array_equal(got[serial], ref32[serial])
    assert np.abs(got - ref64).max() <= 1e-5 * np.abs(ref64).max()
    return lens, ip_s


def test_more_than_two_to_the_31_entries(graph26):
    import torch
    n, a, deg, d = graph26
    ip, ix = a.indptr, a.indices
    assert a.nnz > 2 ** 31 and ip.dtype == torch.int64
    assert int(ip[0]) == 0 and int(ip[-1]) == a.nnz and bool((ip[1:] > ip[:-1]).all())      # +I: no empty row
    assert int((ip > 2 ** 31).sum()) > 500000                                                # rows that start beyond int32
    for k in range(0, a.nnz, 1 << 30):
        assert int(ix[k:k + (1 << 30)].min()) >= 0 and int(ix[k:k + (1 << 30)].max()) < n
    # simple undirected graph: as many entries in column v as in row v
    col_counts = torch.zeros(n, dtype=torch.int64, device='cuda')
    for k in range(0, a.nnz, 1 << 29):
        col_counts += torch.bincount(ix[k:k + (1 << 29)].to(torch.int64), minlength=n)
    assert torch.equal(col_counts, ip[1:] - ip[:-1])
    del col_counts
    assert torch.equal(deg, (ip[1:] - ip[:-1]).to(torch.float64))
    # strictly ascending columns inside every row and one diagonal entry per row, checked on a window at
    # either side of entry 2^31 and on the last rows of the matrix
    for r0 in (0, int(torch.searchsorted(ip, torch.tensor([2 ** 31], device='cuda'))[0]) - 1000, n - 200000):
        r1 = r0 + 200000
        k0, k1 = int(ip[r0]), int(ip[r1])
        rows = torch.repeat_interleave(torch.arange(r0, r1, device='cuda'), ip[r0 + 1:r1 + 1] - ip[r0:r1])
        seg = ix[k0:k1].to(torch.int64)
        same = rows[1:] == rows[:-1]
        assert bool((seg[1:][same] > seg[:-1][same]).all())
        assert int((rows == seg).sum()) == r1 - r0


@pytest.mark.parametrize('f, serial_up_to', [(8, 256), (128, 1024)])
def test_products_on_the_scale_26_graph(graph26, oracle, f, serial_up_to):
    import torch
    from gcn_amd import sparse
    n, a, deg, d = graph26
    b = sparse.synthetic_dense(n, f, seed=7)
    c = sparse.spmm(a, b)
    # 1. exact linearity
    b2 = b * 2
    c2 = sparse.spmm(a, b2)
    del b2
    assert _equals_twice(c2, c)
    del c2
    # 2. sampled rows: 2,000 at random, 500 more among the rows stored beyond entry 2^31, the 4 longest, the ends
    rng = np.random.default_rng(26)
    lens_all = a.indptr[1:] - a.indptr[:-1]
    longest = torch.topk(lens_all, 4).indices.cpu().numpy()
    first_high = int(torch.searchsorted(a.indptr, torch.tensor([2 ** 31], device='cuda'))[0])
    assert 0 < first_high < n - 1000
    sample = np.unique(np.concatenate([rng.integers(0, n, 2000), rng.integers(first_high, n, 500), longest,
                                       [0, first_high - 1, first_high, n - 1]])).astype(np.int64)
    lens, starts = _sampled_rows_against_the_oracle(oracle, a, b, c, sample, serial_up_to)
    assert (starts > 2 ** 31).sum() >= 500 and (lens > serial_up_to).sum() >= 4
    # 3. symmetry of the operator: <y, A x> == <A y, x>
    y = sparse.synthetic_dense(n, f, seed=11)
    cy = sparse.spmm(a, y)
    lhs, rhs = _dot64(y, c), _dot64(cy, b)
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    del y, cy, b, c
    torch.cuda.empty_cache()
    # 4. A . 1 = the row sums of the values, every column the same bits
    if f == 8:
        ones = torch.ones((n, 8), dtype=torch.float32, device='cuda')
        s = sparse.spmm(a, ones)
        assert bool((s == s[:, :1]).all())
        row_sum = sparse.row_sums(a.indptr, a.values.double())
        assert float((s[:, 0].double() - row_sum).abs().max()) <= 1e-5 * float(row_sum.abs().max())


def test_last_of_eight_row_blocks_and_its_compact_operand_product(graph26):
    import torch
    from gcn_amd import sparse, dist as gdist
    n, a, deg, d = graph26
    world, rank = 8, 7
    r0, r1 = gdist.row_block(n, rank, world)
    k0, k1 = int(a.indptr[r0]), int(a.indptr[r1])
    assert k0 > 2 ** 31 - 2 ** 29                    # the block sits in the upper eighth of the entries
    # the block generated on its own equals the slice of the whole graph bit for bit
    indptr, indices = sparse.rmat_csr_rows(SCALE, r0, r1, EDGE_FACTOR, seed=42, self_loops=True)
    assert torch.equal(indptr, a.indptr[r0:r1 + 1] - k0)
    assert torch.equal(indices, a.indices[k0:k1])
    # its values from the degree vector of the whole graph, as dist.rmat_shard scales them
    whole = torch.empty(n + 1, dtype=torch.int64, device='cuda')
    whole[:r0] = 0
    whole[r0:r1 + 1] = indptr
    whole[r1 + 1:] = int(indices.numel())
    vals = sparse.normalize_values(whole, indices, None, d, 'sym', want64=False)[1]
    del whole
    assert torch.equal(vals, a.values[k0:k1])
    # product through the exchange plan's compact operand: the operand holds only the rows of B the block
    # references, in column order, and the block's column indices are positions in it
    local = sparse.DeviceCSR(r1 - r0, n, indptr, indices, vals)
    shard = gdist.RowShardedCSR(n, n, r0, r1, local, 0, 1)
    needed = torch.unique(indices.to(torch.int64))
    assert 0.2 * n < needed.numel() < 0.6 * n         # a skewed graph: a block references a fraction of the rows
    plan = gdist.HaloPlan(shard, operand_rows=needed)
    assert plan.n_first == needed.numel() and plan.n_remote == 0 and plan.compact.n_cols == needed.numel()
    for f in (8, 128):
        b = sparse.synthetic_dense(n, f, seed=7)
        c_rows = sparse.spmm(a, b)[r0:r1].clone()
        operand = b[needed].contiguous()
        del b
        got = plan.multiply(operand, plan.exchange(operand))
        assert torch.equal(got, c_rows)
        del got, c_rows, operand
        plan.drop_buffers()
        torch.cuda.empty_cache()
