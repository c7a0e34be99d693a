"""GPU, 2 ranks sharing cuda:0 over gloo: the row-sharded product with the HIP kernels.  Both
exchange forms (all-gather of B, exchange of the referenced rows only) must reproduce the
single-GPU product bit for bit, because no row's accumulation order changes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, result_dir, scale, f):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gcn_amd import dist as gdist, sparse
        n = 1 << scale
        indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
        d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
        vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
        b = sparse.synthetic_dense(n, f, seed=7)
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, rank, world)
        b_local = b[shard.row_begin:shard.row_end].clone()
        c_gather = gdist.AllGatherSpmm(shard, f)(b_local, relu=True).clone()
        halo = gdist.HaloSpmm(shard, f)
        c_halo = halo(b_local, relu=True)
        assert torch.equal(c_gather, c_halo)
        assert halo.n_needed < n                 # most low-degree vertices are not referenced by a shard
        np.save(os.path.join(result_dir, 'c_%d.npy' % rank), c_halo.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('f', [16, 256])
def test_two_rank_product_equals_single_gpu(f, tmp_path):
    from gcn_amd import sparse
    scale, world = 14, 2
    port = 29700 + (os.getpid() % 1000) + f % 7
    mp.start_processes(_worker, args=(world, port, str(tmp_path), scale, f), nprocs=world, join=True,
                       start_method='spawn')
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    ref = sparse.spmm(a, sparse.synthetic_dense(n, f, seed=7), relu=True).cpu().numpy()
    got = np.concatenate([np.load(os.path.join(str(tmp_path), 'c_%d.npy' % r)) for r in range(world)])
    # rows up to the split threshold are summed serially on both sides: identical bits; a longer row is
    # cut at the same places too, because a shard keeps whole rows
    assert np.array_equal(got, ref)
