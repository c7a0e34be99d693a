"""gcn_spmm_csr_f32_noalloc -- the product entry point that allocates nothing -- through the C ABI,
against gcn_spmm_csr_f32 on the same inputs: the same bits, on two calls in a row of which the second is
the smaller one (the library's array keeps the first call's partial rows), then the size limit.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _skewed(rng, n, hubs, hub_deg, base_deg):
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


def _csr(n, ip, ix, v):
    import torch
    from gcn_amd import sparse
    return sparse.DeviceCSR(n, n, torch.as_tensor(ip).cuda(), torch.as_tensor(ix).cuda(),
                            torch.as_tensor(v).cuda())


def _device_free_bytes():
    """Free device memory as the HIP runtime itself reports it (hipMemGetInfo through ctypes): it sees every
    allocation of the process, those made inside libgcn_hip.so included, which torch's counters do not."""
    hip = ctypes.CDLL('libamdhip64.so')
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_no_alloc_entry_point_equals_the_workspace_one(oracle):
    import torch
    from gcn_amd import _lib, sparse
    # first call: scale-17 R-MAT A_hat, 256 columns -- thousands of long-row chunks in the scratch array
    n1 = 1 << 17
    ip, ix = sparse.rmat_csr(17, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(ip, None), 'sym')
    v = sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1]
    big = sparse.DeviceCSR(n1, n1, ip, ix, v)
    # second call: a 6000-node graph with four hub rows, 35 columns (the scalar-lane path)
    rng = np.random.RandomState(12)
    n2 = 6000
    ip2, ix2 = _skewed(rng, n2, {0, 17, 2999, 5999}, hub_deg=5000, base_deg=6)
    v2 = rng.uniform(-1, 1, ix2.size).astype(np.float32)
    small = _csr(n2, ip2, ix2, v2)
    assert big.plan.n_chunks > 1000 and small.plan.n_chunks > 0            # plans exist; both use the scratch

    need = ctypes.c_size_t()
    _lib.call('gcn_spmm_plan_workspace_bytes', big.plan.handle, 256, ctypes.byref(need))
    assert 0 < need.value <= sparse.NOALLOC_SCRATCH_BYTES

    b1 = sparse.synthetic_dense(n1, 256, seed=3)
    b2 = sparse.synthetic_dense(n2, 35, seed=4)
    ref1 = sparse.spmm(big, b1)
    ref1_relu = sparse.spmm(big, b1, relu=True)
    ref2 = sparse.spmm(small, b2)
    out1 = torch.full_like(ref1, float('nan'))
    out2 = torch.full_like(ref2, float('nan'))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before = _device_free_bytes()
    sparse.spmm_noalloc(big, b1, out1)
    sparse.spmm_noalloc(small, b2, out2)                 # smaller input, right after: stale partial rows beyond it
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert _device_free_bytes() == free_before           # nor did the library allocate behind torch's back
    assert torch.equal(out1, ref1)
    assert torch.equal(out2, ref2)
    # and the other way round, with the ReLU epilogue, into a column slice of a wider matrix
    wide = torch.full((n1, 300), -5.0, device='cuda')
    sparse.spmm_noalloc(small, b2, out2.fill_(float('nan')))
    sparse.spmm_noalloc(big, b1, wide[:, 4:260], relu=True)
    assert torch.equal(out2, ref2)
    assert torch.equal(wide[:, 4:260], ref1_relu)
    assert bool((wide[:, :4] == -5.0).all()) and bool((wide[:, 260:] == -5.0).all())
    # against the oracle too (rows up to the split threshold bit for bit)
    o = oracle.spmm_csr(ip2, ix2, v2, b2.cpu().numpy())
    short = np.diff(ip2) <= small.plan.split_nnz
    assert np.array_equal(out2.cpu().numpy()[short], o[short])

    # a matrix without long rows uses no scratch at all
    fx = np.load(__import__('os').path.join(__import__('conftest').GOLDEN, 'cora.npz'))
    nc = int(fx['n'])
    cora = _csr(nc, fx['ai_indptr'].astype(np.int64), fx['ai_indices'], fx['hat_values'].astype(np.float32))
    assert cora.plan.n_chunks == 0
    bc = sparse.synthetic_dense(nc, 16, seed=5)
    outc = torch.empty((nc, 16), device='cuda')
    assert torch.equal(sparse.spmm_noalloc(cora, bc, outc), sparse.spmm(cora, bc))

    # beyond the array's 32 MiB the call is refused with a clear message and writes nothing
    f_big = 1
    while True:
        _lib.call('gcn_spmm_plan_workspace_bytes', small.plan.handle, f_big, ctypes.byref(need))
        if need.value > sparse.NOALLOC_SCRATCH_BYTES:
            break
        f_big *= 2
    b_big = torch.zeros((n2, f_big), device='cuda')
    out_big = torch.full((n2, f_big), -3.0, device='cuda')
    with pytest.raises(ValueError, match='no-allocation'):
        sparse.spmm_noalloc(small, b_big, out_big)
    torch.cuda.synchronize()
    assert bool((out_big == -3.0).all())
    # the plan must exist before the call (its creation allocates)
    fresh = _csr(n2, ip2, ix2, v2)
    with pytest.raises(ValueError, match='plan'):
        sparse.spmm_noalloc(fresh, b2, out2)


def test_no_alloc_entry_point_under_stream_capture():
    """The use the entry point is documented for: recorded once in a HIP graph on a matrix whose long rows
    go through the library's scratch array, replayed with new contents of B.  An allocation, a free or a
    synchronisation inside the call would make the capture fail; every replay must give the bits of the
    eager product."""
    import torch
    from gcn_amd import sparse
    n = 1 << 17
    ip, ix = sparse.rmat_csr(17, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(ip, None), 'sym')
    v = sparse.normalize_values(ip, ix, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, ip, ix, v)
    assert a.plan.n_chunks > 1000                                   # made before the capture: plan creation allocates
    f = 64
    b_static = sparse.synthetic_dense(n, f, seed=1)
    out_static = torch.full((n, f), float('nan'), device='cuda')
    sparse.spmm_noalloc(a, b_static, out_static)                    # first call outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sparse.spmm_noalloc(a, b_static, out_static, relu=True)
    for seed in (5, 6):
        b_new = sparse.synthetic_dense(n, f, seed=seed)
        expected = sparse.spmm(a, b_new, relu=True)
        b_static.copy_(b_new)
        out_static.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_static, expected)
    del graph
