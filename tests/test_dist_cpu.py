"""CPU, world_size 2 over gloo: the row-partitioned product (gcn_amd/dist.py) -- contiguous row
blocks, all-gather of the dense operand, local product -- gives exactly the single-process rows.
The HIP product cannot run here, so the test injects the CPU oracle as the local multiply; the
partition, the collective and the buffer handling are the code under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


class _HostCSR(object):
    """Stand-in with DeviceCSR's fields, for host tensors."""

    def __init__(self, n_rows, n_cols, indptr, indices, values):
        self.n_rows, self.n_cols = n_rows, n_cols
        self.indptr, self.indices, self.values = indptr, indices, values
        self.device = indptr.device

    @property
    def nnz(self):
        return int(self.indices.numel())


def _oracle_spmm(a, b, relu=False, out=None, wide_split=False, out_dropout=None, add=None):
    import gcn_oracle
    c = gcn_oracle.spmm_csr(a.indptr.numpy(), a.indices.numpy(), a.values.numpy(), b.numpy(), relu=relu and add is None)
    if add is not None:                         # the addend after the row's own sum, then the activation
        c = c + add.numpy()
        if relu:
            c = np.maximum(c, 0)
    if out_dropout is not None:                 # decision ids as the HIP store computes them
        d = out_dropout
        pitch = c.shape[1] if d.pitch is None else d.pitch
        ids = d.index_offset + np.arange(c.shape[0])[:, None] * pitch + np.arange(c.shape[1])[None, :]
        keep = gcn_oracle.keep_mask(gcn_oracle.stream_key(d.seed, d.stream_id), ids, d.keep_prob)
        c = (c / np.float32(d.keep_prob)) * keep.astype(np.float32)
    out.copy_(torch.from_numpy(c))
    return out


def _worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gcn_amd import dist as gdist, sparse
        import gcn_oracle
        fx = load_golden('cora')
        n = int(fx['n'])
        pad = (-n) % world                       # equal blocks: pad with empty rows / unused columns
        n_pad = n + pad
        indptr = np.concatenate([fx['ai_indptr'].astype(np.int64), np.full(pad, fx['ai_indptr'][-1], np.int64)])
        full_indptr = torch.from_numpy(indptr)
        indices = torch.from_numpy(fx['ai_indices'].astype(np.int32))
        values = torch.from_numpy(fx['hat_values'].astype(np.float32))
        r0, r1 = gdist.row_block(n_pad, rank, world)
        k0, k1 = int(full_indptr[r0]), int(full_indptr[r1])
        local = _HostCSR(r1 - r0, n_pad, full_indptr[r0:r1 + 1] - k0, indices[k0:k1].clone(), values[k0:k1].clone())
        shard = gdist.RowShardedCSR(n_pad, n_pad, r0, r1, local, rank, world)
        f = 16
        b = np.zeros((n_pad, f), dtype=np.float32)
        b[:n] = gcn_oracle.synthetic_dense(n, f, seed=13)
        op = gdist.AllGatherSpmm(shard, f, spmm=_oracle_spmm)
        c_local = op(torch.from_numpy(b[r0:r1].copy()), relu=True)
        assert torch.equal(op.b_full, torch.from_numpy(b))             # the all-gather rebuilt B exactly
        np.save(os.path.join(result_dir, 'c_%d.npy' % rank), c_local.numpy())
        # the exchange that ships only the referenced rows of B gives the same bits
        halo = gdist.HaloSpmm(shard, f, spmm=_oracle_spmm, csr_factory=_HostCSR)
        c_halo = halo(torch.from_numpy(b[r0:r1].copy()), relu=True)
        assert torch.equal(c_halo, c_local)
        # own rows are read in place; only other ranks' rows are received (none from oneself)
        assert sum(halo.recv_splits) == halo.n_remote and halo.recv_splits[rank] == 0
        assert halo.plan.n_first == r1 - r0 and halo.compact.n_cols == halo.plan.n_first + halo.n_remote
        np.save(os.path.join(result_dir, 'needed_%d.npy' % rank), np.array([halo.n_remote]))
        # pipelined over ranges of feature columns (uneven ranges, two buffer slots): still the same bits
        piped = gdist.HaloSpmm(shard, f, spmm=_oracle_spmm, plan=halo.plan, chunks=3)
        assert piped.ranges == [(0, 4), (4, 8), (8, 16)]
        assert torch.equal(piped(torch.from_numpy(b[r0:r1].copy()), relu=True), c_local)
        # dropout in the product's store: a rank draws its row block's decisions of the unsplit matrix,
        # whole or range by range
        drop = sparse.OutDropout(0.5, 11, 3, index_offset=r0 * f)
        want, _ = gcn_oracle.dense_dropout(c_local.numpy(), 0.5, gcn_oracle.stream_key(11, 3), index_offset=r0 * f)
        for ranges in ([(0, f)], [(0, 4), (4, 8), (8, 16)]):
            got = gdist.halo_product(halo.plan, torch.from_numpy(b[r0:r1].copy()), torch.empty_like(c_local), True,
                                     ranges, _oracle_spmm, out_dropout=drop)
            assert np.array_equal(got.numpy(), want)
        # own rows first: the entries that read the rank's own block are summed before the others (the part
        # that can run while the exchange is on the links).  Bit for bit what the CPU loop gives for that order
        # of additions -- own entries, then the other entries, then the two sums added --, whole or range by
        # range, with the activation and the store dropout applied once at the end; and equal to the in-order
        # product to rounding.
        ip, ix, v = local.indptr.numpy(), local.indices.numpy(), local.values.numpy()
        own = (ix >= r0) & (ix < r1)
        row_of = np.repeat(np.arange(r1 - r0), np.diff(ip))

        def part(keep):
            cnt = np.bincount(row_of[keep], minlength=r1 - r0)
            return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), ix[keep], v[keep]
        first = gcn_oracle.spmm_csr(*part(own), b)
        both = np.maximum(gcn_oracle.spmm_csr(*part(~own), b) + first, 0)
        assert np.abs(both - c_local.numpy()).max() <= 1e-6 * np.abs(c_local.numpy()).max()
        own_plan = gdist.HaloPlan(shard, csr_factory=_HostCSR, own_first=True)
        assert own_plan.own_first and own_plan.compact is None
        assert own_plan.own.nnz == int(own.sum()) and own_plan.remote.nnz == int((~own).sum())
        assert own_plan.own.n_cols == r1 - r0 and own_plan.remote.n_cols == own_plan.n_remote == halo.n_remote
        want_dropped, _ = gcn_oracle.dense_dropout(both, 0.5, gcn_oracle.stream_key(11, 3), index_offset=r0 * f)
        for ranges in ([(0, f)], [(0, 4), (4, 8), (8, 16)]):
            got = gdist.halo_product(own_plan, torch.from_numpy(b[r0:r1].copy()), torch.empty_like(c_local), True,
                                     ranges, _oracle_spmm)
            assert np.array_equal(got.numpy(), both)
            got = gdist.halo_product(own_plan, torch.from_numpy(b[r0:r1].copy()), torch.empty_like(c_local), True,
                                     ranges, _oracle_spmm, out_dropout=drop)
            assert np.array_equal(got.numpy(), want_dropped)
        assert np.array_equal(own_plan.multiply(torch.from_numpy(b[r0:r1].copy()),
                                                own_plan.exchange(torch.from_numpy(b[r0:r1].copy())), relu=True,
                                                out=torch.empty_like(c_local), spmm=_oracle_spmm).numpy(), both)
        auto = gdist.HaloPlan(shard, csr_factory=_HostCSR, own_first='auto')
        assert auto.own_first == (own.sum() >= 0.2 * own.size)
        # the pipeline is picked by measurement; a rigged clock makes the choice predictable, the
        # MAX over ranks makes it the same everywhere, and the result does not change
        ticks = iter(range(1000))
        fake = {'t': 0.0}

        def clock():
            fake['t'] += next(ticks) * (1.0 + rank)          # later candidates look slower, on every rank
            return fake['t']
        timings = piped.autotune(torch.from_numpy(b[r0:r1].copy()), reps=1, clock=clock)
        assert len(timings) >= 3 and piped.ranges == timings[0][0] == [(0, f)]
        assert torch.equal(piped(torch.from_numpy(b[r0:r1].copy()), relu=True), c_local)
        # the MAX-over-ranks timing reduction bench.py uses
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == float(world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])        # 8: the rank count of the full node
def test_row_sharded_product_equals_single_process(world, oracle, tmp_path):
    port = 29500 + (os.getpid() % 2000) + world
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    fx = load_golden('cora')
    n = int(fx['n'])
    b = oracle.synthetic_dense(n, 16, seed=13)
    ref = oracle.spmm_csr(fx['ai_indptr'].astype(np.int64), fx['ai_indices'], fx['hat_values'].astype(np.float32), b,
                          relu=True)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), 'c_%d.npy' % r)) for r in range(world)])[:n]
    assert np.array_equal(got, ref)          # same per-row order on every rank: bit-identical
    for r in range(world):                   # a shard references only part of the columns
        assert int(np.load(os.path.join(str(tmp_path), 'needed_%d.npy' % r))[0]) < n


def test_column_chunks_cover_every_column_once():
    sys.path.insert(0, ROOT)
    from gcn_amd import dist as gdist
    for f in (1, 6, 7, 16, 100, 256, 513):
        for chunks in (1, 2, 4, 7):
            for progressive in (True, False):
                ranges = gdist.column_chunks(f, chunks, progressive=progressive)
                assert ranges[0][0] == 0 and ranges[-1][1] == f and len(ranges) <= max(chunks, 1)
                assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
                assert all(lo % 4 == 0 and hi > lo for lo, hi in ranges)
    assert gdist.column_chunks(256, 4) == [(0, 32), (32, 64), (64, 128), (128, 256)]
    assert gdist.column_chunks(256, 4, progressive=False) == [(0, 64), (64, 128), (128, 192), (192, 256)]


def test_row_block_requires_equal_blocks():
    sys.path.insert(0, ROOT)
    from gcn_amd import dist as gdist
    assert gdist.row_block(8, 1, 4) == (2, 4)
    with pytest.raises(ValueError):
        gdist.row_block(10, 0, 4)
