"""GPU, 2 ranks sharing cuda:0 over gloo: the row-sharded product with the HIP kernels.  Both
exchange forms (all-gather of B, exchange of the referenced rows only) must reproduce the
single-GPU product bit for bit, because no row's accumulation order changes."""
import faulthandler
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
    faulthandler.dump_traceback_later(90, exit=True)      # a stuck collective names itself and ends the rank
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
        assert 0 < halo.n_remote < n // 2        # most low-degree vertices of the other blocks are not referenced
        assert halo.recv_splits[rank] == 0       # own rows are read where they lie, never exchanged
        piped = gdist.HaloSpmm(shard, f, plan=halo.plan, chunks=4)
        assert torch.equal(piped(b_local, relu=True), c_halo)
        np.save(os.path.join(result_dir, 'c_%d.npy' % rank), c_halo.cpu().numpy())
        # own rows first (the part of the product that hides the exchange): one range and three, twice each
        own = gdist.HaloSpmm(shard, f, own_first=True)
        assert own.plan.own_first and own.plan.own.nnz + own.plan.remote.nnz == shard.local.nnz
        c_own = own(b_local, relu=True).clone()
        assert torch.equal(own(b_local, relu=True), c_own)
        own3 = gdist.HaloSpmm(shard, f, plan=own.plan, chunks=3)
        assert len(own3.ranges) == 3 and torch.equal(own3(b_local, relu=True), c_own)
        np.save(os.path.join(result_dir, 'own_%d.npy' % rank), c_own.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('f', [16, 256])
def test_two_rank_product_equals_single_gpu(f, tmp_path, oracle):
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
    # and the two ranks against the CPU loop itself, not only through the one-GPU product
    ip, ix, v = indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy()
    bn = oracle.synthetic_dense(n, f, seed=7)
    cpu = oracle.spmm_csr(ip, ix, v, bn, relu=True, threads=8)
    short = np.diff(ip) <= (256 if f < 128 else 1024)
    assert short.sum() > 0.95 * n
    assert np.array_equal(got[short], cpu[short])
    exact = np.maximum(oracle.spmm_csr_f64(ip, ix, v, bn), 0)
    assert np.abs(got - exact).max() <= 4e-6 * max(1.0, np.abs(exact).max())
    # own rows first: another order of the same additions -- the same product to rounding, and on the rows
    # short enough to be summed serially exactly what the CPU loop gives for that order
    own = np.concatenate([np.load(os.path.join(str(tmp_path), 'own_%d.npy' % r)) for r in range(world)])
    assert np.abs(own - exact).max() <= 4e-6 * max(1.0, np.abs(exact).max())
    assert not np.array_equal(own, got)
    per = n // world
    block = np.repeat(np.arange(n), np.diff(ip)) // per
    mine = (ix // per) == block

    def part(keep):
        cnt = np.bincount(np.repeat(np.arange(n), np.diff(ip))[keep], minlength=n)
        return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), ix[keep], v[keep]
    cpu_own = np.maximum(oracle.spmm_csr(*part(~mine), bn, threads=8) + oracle.spmm_csr(*part(mine), bn, threads=8), 0)
    assert np.array_equal(own[short], cpu_own[short])


def test_row_range_generation_equals_the_slice_of_the_whole_graph():
    """Per-rank generation (gcn_rmat_csr_rows, gcn_fill_synthetic_range_f32) against slices of the whole
    graph and the whole dense operand, bit for bit, for every rank of 2-, 3- and 8-rank splits at scale 18."""
    from gcn_amd import sparse
    scale, f = 18, 24
    n = 1 << scale
    for scramble in (True, False):
        indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True, scramble=scramble)
        for world in (2, 3, 8):
            edges = [n * r // world for r in range(world + 1)]
            for r0, r1 in zip(edges, edges[1:]):
                ip, ix = sparse.rmat_csr_rows(scale, r0, r1, 16, seed=42, self_loops=True, scramble=scramble)
                k0, k1 = int(indptr[r0]), int(indptr[r1])
                assert torch.equal(ip, indptr[r0:r1 + 1] - k0)
                assert torch.equal(ix, indices[k0:k1])
    plain_ip, plain_ix = sparse.rmat_csr_rows(scale, 1000, 3000, 16, seed=42, self_loops=False)
    whole_ip, whole_ix = sparse.rmat_csr(scale, 16, seed=42, self_loops=False)
    assert torch.equal(plain_ix, whole_ix[int(whole_ip[1000]):int(whole_ip[3000])])
    empty_ip, empty_ix = sparse.rmat_csr_rows(scale, 77, 77, 16, seed=42, self_loops=True)
    assert empty_ip.tolist() == [0] and empty_ix.numel() == 0
    b = sparse.synthetic_dense(n, f, seed=7)
    assert torch.equal(sparse.synthetic_dense(5000, f, seed=7, first_row=12345), b[12345:17345])


def _rmat_shard_worker(rank, world, port, result_dir, scale):
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gcn_amd import dist as gdist, sparse
        n = 1 << scale
        shard = gdist.rmat_shard(scale, 16, rank, world, seed=42)
        indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
        d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
        vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
        ref = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, rank, world)
        assert (shard.row_begin, shard.row_end) == (ref.row_begin, ref.row_end)
        assert torch.equal(shard.local.indptr, ref.local.indptr)
        assert torch.equal(shard.local.indices, ref.local.indices)
        assert torch.equal(shard.local.values, ref.local.values)
        f = 64
        b_local = sparse.synthetic_dense(shard.row_end - shard.row_begin, f, seed=7, first_row=shard.row_begin)
        c = gdist.HaloSpmm(shard, f)(b_local)
        whole = sparse.spmm(sparse.DeviceCSR(n, n, indptr, indices, vals), sparse.synthetic_dense(n, f, seed=7))
        assert torch.equal(c, whole[shard.row_begin:shard.row_end])
        open(os.path.join(result_dir, 'ok_%d' % rank), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_ranks_generate_their_own_shards(tmp_path):
    """2 ranks on one GPU at scale 18: each generates only its row block of A_hat (degrees exchanged by
    all-gather) and its rows of B; shard and product equal the slices of the single-process ones."""
    world = 2
    port = 29500 + (os.getpid() % 1000)
    mp.start_processes(_rmat_shard_worker, args=(world, port, str(tmp_path), 18), nprocs=world, join=True,
                       start_method='spawn')
    assert all((tmp_path / ('ok_%d' % r)).exists() for r in range(world))


def _single_rank_rccl_worker(rank, port, result_dir):
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from gcn_amd import dist as gdist, sparse
        scale, f = 13, 256
        n = 1 << scale
        indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
        d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
        vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
        b = sparse.synthetic_dense(n, f, seed=7)
        ref = sparse.spmm(sparse.DeviceCSR(n, n, indptr, indices, vals), b, relu=True)
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
        piped = gdist.HaloSpmm(shard, f, chunks=4)
        assert piped.plan._collective
        for _ in range(3):                         # buffers are reused across calls
            got = piped(b, relu=True)
            assert torch.equal(got, ref)
        timings = piped.autotune(b)                # every candidate split runs over RCCL; the fastest is kept
        assert len(timings) == len(gdist.HaloSpmm.CANDIDATES) and piped.ranges in [r for r, _ in timings]
        assert torch.equal(piped(b, relu=True), ref)
        gathered = gdist.AllGatherSpmm(shard, f)(b, relu=True)
        assert torch.equal(gathered, ref)
        # own rows first over RCCL: with one rank every entry reads an own row, the part added afterwards is
        # empty (activation in its store all the same) and the bits are the in-order ones
        own = gdist.HaloSpmm(shard, f, chunks=4, own_first=True)
        assert own.plan.own_first and own.plan.remote.nnz == 0 and own.plan.own.nnz == shard.local.nnz
        for _ in range(2):
            assert torch.equal(own(b, relu=True), ref)
        t = torch.ones(4, device='cuda')
        gdist.all_reduce_sum(t)
        assert float(t.sum()) == 4.0
        open(os.path.join(result_dir, 'ok'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_exchange_in_flight_over_rccl(tmp_path):
    """The RCCL form of the exchange (asynchronous all-to-all on RCCL's stream, the product of the
    previous column range on the caller's stream) on a 1-rank communicator: the only RCCL world one
    GPU allows, and the same calls, stream ordering and buffer reuse as N ranks."""
    port = 29600 + (os.getpid() % 1000)
    mp.start_processes(_single_rank_rccl_worker, args=(port, str(tmp_path)), nprocs=1, join=True, start_method='spawn')
    assert (tmp_path / 'ok').exists()


def _abi_comm_worker(rank, result_dir):
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path.insert(0, ROOT)
    import ctypes
    torch.cuda.set_device(0)
    from gcn_amd import _lib, sparse
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    uid = ctypes.create_string_buffer(128)
    _lib.call('gcn_comm_unique_id', uid)
    comm = ctypes.c_void_p()
    _lib.call('gcn_comm_init_rank', ctypes.byref(comm), 1, uid, 0)
    assert comm.value
    n, f = 5000, 64
    b = sparse.synthetic_dense(n, f, seed=7)
    gathered = torch.full((n, f), float('nan'), device='cuda')
    _lib.call('gcn_allgather_rows_f32', comm, b.data_ptr(), gathered.data_ptr(), n, f, stream)
    torch.cuda.synchronize()
    assert torch.equal(gathered, b)
    # the referenced-rows exchange with oneself: 700 packed rows out, the same 700 rows in
    rows = torch.arange(0, 4900, 7, device='cuda')
    send = torch.empty((rows.numel(), f), device='cuda')
    sparse.gather_rows(b, rows, send)
    recv = torch.full((rows.numel(), f), float('nan'), device='cuda')
    counts = (ctypes.c_int64 * 1)(rows.numel())
    _lib.call('gcn_exchange_rows_f32', comm, send.data_ptr(), counts, recv.data_ptr(), counts, f, stream)
    torch.cuda.synchronize()
    assert torch.equal(recv, b[rows])
    # zero rows and bad arguments
    zero = (ctypes.c_int64 * 1)(0)
    _lib.call('gcn_exchange_rows_f32', comm, None, zero, None, zero, f, stream)
    with pytest.raises(ValueError):
        _lib.call('gcn_allgather_rows_f32', None, b.data_ptr(), gathered.data_ptr(), n, f, stream)
    _lib.call('gcn_comm_destroy', comm)
    open(os.path.join(result_dir, 'ok'), 'w').write('ok')


def test_rccl_entry_points_of_the_c_abi(tmp_path):
    """gcn_comm_* / gcn_allgather_rows_f32 / gcn_exchange_rows_f32 through ctypes alone (no
    torch.distributed) on a 1-rank communicator, the only RCCL world one GPU allows."""
    mp.start_processes(_abi_comm_worker, args=(str(tmp_path),), nprocs=1, join=True, start_method='spawn')
    assert (tmp_path / 'ok').exists()


def _train_worker(rank, world, port, result_dir, scale, steps, aggregate_first, own_first=False, widths=(32, 128),
                  tag='', **model_options):
    if world > 1:
        faulthandler.dump_traceback_later(90 if world <= 2 else 150, exit=True)
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gcn_amd import dist as gdist, sparse
        from gcn_amd.dist_train import ShardedGCN
        n, f_in, classes, hidden = 1 << scale, widths[0], 8, widths[1]
        indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
        d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
        vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, rank, world)
        r0, r1 = shard.row_begin, shard.row_end
        x = sparse.synthetic_dense(n, f_in, seed=7)[r0:r1].contiguous()
        gen = torch.Generator().manual_seed(5)
        cls = torch.randint(0, classes, (n,), generator=gen)
        labels = torch.nn.functional.one_hot(cls, classes).to(torch.int32)[r0:r1].cuda()
        mask = (torch.rand(n, generator=gen) < 0.05).to(torch.int32)[r0:r1].cuda()
        model = ShardedGCN(shard, f_in, [hidden], classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.5, seed=9,
                           aggregate_first=aggregate_first, own_first=own_first, **model_options)
        if own_first != 'auto':                     # 'auto' decides by the share of entries that read own rows
            assert model.plan.own_first == (bool(own_first) and world > 1)
        losses = [float(model.train_step(x, labels, mask)) for _ in range(steps)]
        if rank == 0:
            np.save(os.path.join(result_dir, 'loss_w%d%s.npy' % (world, tag)), np.array(losses))
            np.save(os.path.join(result_dir, 'w0_w%d%s.npy' % (world, tag)), model.weights[0].detach().cpu().numpy())
            if (world > 1 and model_options.get('labelled_logits_only') == 'backward' and 128 < hidden <= 256
                    and not own_first):
                hidden_plan = next(iter(model._grad_plans.values()))[1][1].hidden_plan
                assert hidden_plan._packed_gradient.n_rows == hidden_plan.n_first + hidden_plan.n_remote
            if world > 1 and model_options.get('cache_feature_halo', True) and 128 < f_in <= 256 and aggregate_first:
                assert len(model._feature_halos) == 1          # one exchange of feature rows, at the first step
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize('aggregate_first', [False, True])
def test_sharded_training_step_matches_single_gpu(tmp_path, aggregate_first):
    """Forward + backward + Adam on 2 ranks (exchange of referenced rows in both directions, all-reduced
    weight gradients, the unsplit dropout masks) against the same model on 1 rank, with the first layer
    in the reference's order A . (X' . W) and in the order (A . X') . W that needs no backward product."""
    scale, steps = 13, 4
    port = 29800 + (os.getpid() % 1000) + (7 if aggregate_first else 0)
    _train_worker(0, 1, port, str(tmp_path), scale, steps, aggregate_first)
    mp.start_processes(_train_worker, args=(2, port + 1, str(tmp_path), scale, steps, aggregate_first), nprocs=2,
                       join=True, start_method='spawn')
    l1, l2 = np.load(tmp_path / 'loss_w1.npy'), np.load(tmp_path / 'loss_w2.npy')
    assert np.all(np.isfinite(l1)) and l1[-1] < l1[0]
    assert np.allclose(l1, l2, rtol=1e-5)
    w1, w2 = np.load(tmp_path / 'w0_w1.npy'), np.load(tmp_path / 'w0_w2.npy')
    d = np.abs(w1 - w2)
    assert (d <= 1e-5 * np.abs(w1).max()).mean() >= 0.999 and d.max() <= 1e-3 * np.abs(w1).max()


def test_eight_rank_training_step_matches_single_gpu(tmp_path):
    """The rank count of a full node, rehearsed as 8 processes on one GPU over gloo: the default form of the step
    (first layer aggregated first, hidden rows only where a labelled node reads them) with the rule bench.py passes
    for the order of additions (own rows first where the own part holds a fifth of a shard's entries), against 1 rank."""
    scale, steps = 13, 3
    port = 30100 + (os.getpid() % 1000)
    _train_worker(0, 1, port, str(tmp_path), scale, steps, True)
    mp.start_processes(_train_worker, args=(8, port + 1, str(tmp_path), scale, steps, True, 'auto'), nprocs=8,
                       join=True, start_method='spawn')
    l1, l8 = np.load(tmp_path / 'loss_w1.npy'), np.load(tmp_path / 'loss_w8.npy')
    assert np.all(np.isfinite(l1)) and np.allclose(l1, l8, rtol=1e-5)
    w1, w8 = np.load(tmp_path / 'w0_w1.npy'), np.load(tmp_path / 'w0_w8.npy')
    d = np.abs(w1 - w8)
    assert (d <= 1e-5 * np.abs(w1).max()).mean() >= 0.999 and d.max() <= 1e-3 * np.abs(w1).max()


@pytest.mark.parametrize('receptive_field', [True, False])
@pytest.mark.parametrize('own_first', [False, True])
def test_feature_rows_of_other_ranks_kept_across_steps(tmp_path, receptive_field, own_first):
    """Wide features (160 columns) on 2 ranks: the rows of X a shard reads from the other rank are fetched once and
    dropped locally every step, by their global row numbers, into one packed operand.  The first layer's product keeps
    the single-GPU order of additions either way, so with every other product in order too the losses and weights
    equal, bit for bit, those of the step that drops its own rows and exchanges them every step; both follow 1 rank."""
    scale, steps = 13, 3
    port = 30300 + (os.getpid() % 1000) + (11 if receptive_field else 0) + (23 if own_first else 0)
    common = dict(widths=(160, 160), receptive_field=receptive_field)
    _train_worker(0, 1, port, str(tmp_path), scale, steps, True, **common)
    for tag, cache in (('kept', True), ('sent', False)):
        mp.start_processes(_train_worker_kw, args=(2, port + (1 if cache else 2), str(tmp_path), scale, steps, own_first,
                                                   dict(common, tag=tag, cache_feature_halo=cache)),
                           nprocs=2, join=True, start_method='spawn')
    l1 = np.load(tmp_path / 'loss_w1.npy')
    kept, sent = np.load(tmp_path / 'loss_w2kept.npy'), np.load(tmp_path / 'loss_w2sent.npy')
    assert np.all(np.isfinite(l1)) and np.allclose(l1, kept, rtol=1e-5)
    wk, ws = np.load(tmp_path / 'w0_w2kept.npy'), np.load(tmp_path / 'w0_w2sent.npy')
    if not own_first:
        assert np.array_equal(kept, sent) and np.array_equal(wk, ws)
    else:
        assert np.allclose(kept, sent, rtol=1e-5)
    w1 = np.load(tmp_path / 'w0_w1.npy')
    d = np.abs(w1 - wk)
    assert (d <= 1e-5 * np.abs(w1).max()).mean() >= 0.999 and d.max() <= 1e-3 * np.abs(w1).max()


def _train_worker_kw(rank, world, port, result_dir, scale, steps, own_first, options):
    _train_worker(rank, world, port, result_dir, scale, steps, True, own_first, **options)


@pytest.mark.parametrize('widths', [(32, 128), (64, 192)])
def test_reference_order_step_with_labelled_columns_backward_on_two_ranks(tmp_path, widths):
    """The reference's order of operations on 2 ranks in the form the bench line times: every logit formed, the
    backward product of the logits over the columns of labelled nodes only (a smaller support, a smaller exchange),
    the hidden layer's backward product over the columns of their neighbours only -- and, with 192 hidden units,
    on own and received rows of the gradient packed into one operand.  Against 1 rank in the same form."""
    scale, steps = 13, 3
    port = 30500 + (os.getpid() % 1000) + widths[1]
    form = dict(receptive_field=False, labelled_logits_only='backward', widths=widths)
    _train_worker(0, 1, port, str(tmp_path), scale, steps, False, **form)
    mp.start_processes(_train_worker_kw_ref, args=(2, port + 1, str(tmp_path), scale, steps, form), nprocs=2, join=True,
                       start_method='spawn')
    l1, l2 = np.load(tmp_path / 'loss_w1.npy'), np.load(tmp_path / 'loss_w2.npy')
    assert np.all(np.isfinite(l1)) and np.allclose(l1, l2, rtol=1e-5)
    w1, w2 = np.load(tmp_path / 'w0_w1.npy'), np.load(tmp_path / 'w0_w2.npy')
    d = np.abs(w1 - w2)
    assert (d <= 1e-5 * np.abs(w1).max()).mean() >= 0.999 and d.max() <= 1e-3 * np.abs(w1).max()


def _train_worker_kw_ref(rank, world, port, result_dir, scale, steps, options):
    _train_worker(rank, world, port, result_dir, scale, steps, False, False, **options)


@pytest.mark.parametrize('aggregate_first', [False, True])
def test_sharded_training_step_with_own_rows_first(tmp_path, aggregate_first):
    """The same comparison with every plan of the 2-rank model cut into the entries that read own rows (multiplied
    while the exchange runs) and the rest (added as the rows arrive, activation and dropout in that last store)."""
    scale, steps = 13, 4
    port = 29900 + (os.getpid() % 1000) + (7 if aggregate_first else 0)
    _train_worker(0, 1, port, str(tmp_path), scale, steps, aggregate_first)
    mp.start_processes(_train_worker, args=(2, port + 1, str(tmp_path), scale, steps, aggregate_first, True), nprocs=2,
                       join=True, start_method='spawn')
    l1, l2 = np.load(tmp_path / 'loss_w1.npy'), np.load(tmp_path / 'loss_w2.npy')
    assert np.allclose(l1, l2, rtol=1e-5)
    w1, w2 = np.load(tmp_path / 'w0_w1.npy'), np.load(tmp_path / 'w0_w2.npy')
    d = np.abs(w1 - w2)
    assert (d <= 1e-5 * np.abs(w1).max()).mean() >= 0.999 and d.max() <= 1e-3 * np.abs(w1).max()


@pytest.mark.parametrize('aggregate_first', [False, True])
def test_sharded_step_against_a_dense_float64_model(aggregate_first):
    """The sharded model on one rank, dropout off, against the same 2-layer GCN written with dense
    float64 torch operations and autograd: loss trajectory over 4 Adam steps, for both orders of the
    first layer's products."""
    from gcn_amd import dist as gdist, sparse
    from gcn_amd.dist_train import ShardedGCN
    scale, f_in, hidden, classes = 10, 24, 16, 5
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 8, seed=3, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
    x = sparse.synthetic_dense(n, f_in, seed=7)
    gen = torch.Generator().manual_seed(5)
    cls = torch.randint(0, classes, (n,), generator=gen)
    labels = torch.nn.functional.one_hot(cls, classes).to(torch.int32).cuda()
    mask = (torch.rand(n, generator=gen) < 0.2).to(torch.int32).cuda()
    model = ShardedGCN(shard, f_in, [hidden], classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.0, seed=9,
                       aggregate_first=aggregate_first)
    # dense float64 twin
    rows = torch.repeat_interleave(torch.arange(n, device='cuda'), indptr[1:] - indptr[:-1])
    a = torch.zeros((n, n), dtype=torch.float64, device='cuda')
    a[rows, indices.long()] = vals.double()
    w = [p.detach().double().clone().requires_grad_(True) for p in model.weights]
    m = [torch.zeros_like(p) for p in w]
    v = [torch.zeros_like(p) for p in w]
    lab64, mask64 = labels.double(), mask.double()
    for t in range(1, 5):
        h = torch.relu(a @ (x.double() @ w[0]))
        logits = a @ (h @ w[1])
        ce = -(lab64 * torch.log_softmax(logits, dim=1)).sum(dim=1)
        ref_loss = (ce * mask64 / mask64.mean()).mean() + sum(5e-4 * (p * p).sum() / 2 for p in w)
        grads = torch.autograd.grad(ref_loss, w)
        loss = float(model.train_step(x, labels, mask))
        ref_value = float(ref_loss.detach())
        assert abs(loss - ref_value) <= 1e-4 * abs(ref_value), (t, loss, ref_value)
        lr_t = 0.02 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            for p, g, mm, vv in zip(w, grads, m, v):
                mm += 0.1 * (g - mm)
                vv += 0.001 * (g * g - vv)
                p -= lr_t * mm / (vv.sqrt() + 1e-8)


def test_support_restricted_to_labelled_columns_gives_the_same_bits():
    """A . G for a G that is zero outside the labelled rows, with the whole support and with the support
    cut down to the labelled columns (the last layer's backward product): 1/16 of the entries, identical
    bits on every row the full product sums serially, 1e-5 on the chunked hub rows."""
    from gcn_amd import dist as gdist, sparse
    scale = 14
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
    labelled = (torch.arange(n, device='cuda') % 16 == 3)
    g = sparse.synthetic_dense(n, 7, seed=3) * labelled[:, None]
    cut = gdist.shard_with_columns(shard, labelled)
    assert 0 < cut.local.nnz < shard.local.nnz // 8
    full = sparse.spmm(shard.local, g)
    part = sparse.spmm(cut.local, g)
    # rows the full product sums serially: the same terms in the same order, the zeros left out
    serial = (indptr[1:] - indptr[:-1]) <= shard.local.plan.split_nnz
    assert int((~serial).sum()) > 0 and torch.equal(part[serial], full[serial])
    # hub rows are summed in chunks by the full product and (shorter now) serially by the cut one
    assert float((part - full).abs().max()) <= 1e-5 * float(full.abs().max())
    plan = gdist.HaloPlan(cut)
    assert torch.equal(gdist.halo_product(plan, g, torch.empty_like(full)), part)


def test_receptive_field_step_equals_the_full_step():
    """Training steps that form only the hidden rows the labelled nodes read against steps that form all
    of them: the same dropout decisions (drawn by global row), the same loss, the same weights."""
    from gcn_amd import dist as gdist, sparse
    from gcn_amd.dist_train import ShardedGCN
    scale, f_in, hidden, classes = 14, 32, 64, 8
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
    x = sparse.synthetic_dense(n, f_in, seed=7)
    gen = torch.Generator().manual_seed(5)
    labels = torch.nn.functional.one_hot(torch.randint(0, classes, (n,), generator=gen), classes).to(torch.int32).cuda()
    mask = (torch.arange(n) % 64 == 5).to(torch.int32).cuda()
    runs = []
    for receptive in (False, True):
        model = ShardedGCN(shard, f_in, [hidden], classes, dropout=0.5, seed=9, aggregate_first=True,
                           receptive_field=receptive)
        losses = [float(model.train_step(x, labels, mask)) for _ in range(4)]
        runs.append((losses, [w.detach().clone() for w in model.weights]))
        if receptive:
            rows = model._receptive_plans(mask)['rows']
            assert 0 < rows.numel() < n // 2                      # a fraction of the hidden rows was formed
    # the reference's own form of the step -- A.(X'.W) in both layers, the logits of every row -- follows the
    # same loss curve (its first layer is summed in another order: equal to rounding, not to the bit)
    plain = ShardedGCN(shard, f_in, [hidden], classes, dropout=0.5, seed=9, aggregate_first=False,
                       receptive_field=False, labelled_logits_only=False)
    plain_losses = [float(plain.train_step(x, labels, mask)) for _ in range(4)]
    assert np.allclose(plain_losses, runs[0][0], rtol=1e-4)
    (l0, w0), (l1, w1) = runs
    assert np.allclose(l0, l1, rtol=1e-6) and l0[-1] < l0[0]
    for a, b in zip(w0, w1):        # Adam divides by sqrt(v): entries whose gradient is nearly zero amplify rounding
        d, top = (a - b).abs(), float(a.abs().max())
        assert float((d <= 1e-5 * top).float().mean()) >= 0.999 and float(d.max()) <= 1e-3 * top


@pytest.mark.parametrize('reorder', ['degree', ('hubs', 64)])
@pytest.mark.parametrize('form', [dict(aggregate_first=False, receptive_field=False), dict()])
def test_sharded_model_on_a_relabelled_graph_follows_the_plain_one(reorder, form):
    """ShardedGCN(reorder=...): the graph is relabelled once, inputs are permuted on their way in and the
    forward's rows come back in the dataset's order.  With dropout off it is the same model up to the order
    of the additions: same forward, same loss trajectory and weights over 4 Adam steps (1e-5)."""
    from gcn_amd import dist as gdist, sparse
    from gcn_amd.dist_train import ShardedGCN
    scale, f_in, hidden, classes = 12, 32, 16, 5
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 8, seed=3, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
    x = sparse.synthetic_dense(n, f_in, seed=7)
    gen = torch.Generator().manual_seed(5)
    labels = torch.nn.functional.one_hot(torch.randint(0, classes, (n,), generator=gen), classes).to(torch.int32).cuda()
    mask = (torch.rand(n, generator=gen) < 0.2).to(torch.int32).cuda()
    kw = dict(learning_rate=0.02, weight_decay=5e-4, dropout=0.0, seed=9, **form)
    plain = ShardedGCN(shard, f_in, [hidden], classes, **kw)
    moved = ShardedGCN(shard, f_in, [hidden], classes, reorder=reorder, **kw)
    assert not torch.equal(moved.shard.local.indices, shard.local.indices)
    with torch.no_grad():
        a, b = plain.forward(x, train=False), moved.forward(x, train=False)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    for step in range(4):
        la, lb = float(plain.train_step(x, labels, mask)), float(moved.train_step(x, labels, mask))
        assert abs(la - lb) <= 1e-5 * abs(la), (step, la, lb)
    for wa, wb in zip(plain.weights, moved.weights):
        dw = (wa.detach() - wb.detach()).abs()
        assert float((dw <= 1e-5 * wa.detach().abs().max()).float().mean()) >= 0.999 and float(dw.max()) <= 1e-3 * float(wa.detach().abs().max())
    with pytest.raises(ValueError, match='one rank'):
        two = gdist.RowShardedCSR(n, n, 0, n // 2, shard.local, 0, 2)
        ShardedGCN(two, f_in, [hidden], classes, reorder='degree')


@pytest.mark.parametrize('receptive_field,aggregate_first', [(True, True), (False, True), (False, False)])
def test_products_on_packed_operands_change_nothing(receptive_field, aggregate_first):
    """One GPU, 256 input features and hidden units, ShardedGCN(packed_operands=True) (the default) against False.
    Aggregate-first: dropout(X) is written in packed form and gathered packed.  Reference order: the gradient of
    the hidden layer, zero wherever the unit was dropped or inactive, is packed before the backward product.  The
    same decisions, the same values, the same sums: identical losses and weights after 3 Adam steps."""
    from gcn_amd import dist as gdist, sparse
    from gcn_amd.dist_train import ShardedGCN
    scale, f_in, hidden, classes = 13, 256, 256, 8
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    x = sparse.synthetic_dense(n, f_in, seed=7)
    gen = torch.Generator().manual_seed(5)
    cls = torch.randint(0, classes, (n,), generator=gen)
    labels = torch.nn.functional.one_hot(cls, classes).to(torch.int32).cuda()
    mask = (torch.rand(n, generator=gen) < 0.05).to(torch.int32).cuda()
    runs = []
    for packed in (True, False):
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
        model = ShardedGCN(shard, f_in, [hidden], classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.5, seed=9,
                           receptive_field=receptive_field, aggregate_first=aggregate_first, packed_operands=packed)
        losses = [float(model.train_step(x, labels, mask)) for _ in range(3)]
        assert (model._pack_buffer is not None) == (packed and aggregate_first)
        runs.append((losses, [w.detach().clone() for w in model.weights]))
    if aggregate_first:
        assert runs[0][0] == runs[1][0]
        assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    else:
        # the packed gradient of the hidden layer comes from a kernel that adds the 8 products of its small
        # inner dimension in another order than the tiled GEMM: equal to rounding
        assert np.allclose(runs[0][0], runs[1][0], rtol=2e-6)
        for a, b in zip(runs[0][1], runs[1][1]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize('hidden,labelled_share', [(128, 0.05), (192, 0.002)])
def test_all_logits_forward_labelled_columns_backward(hidden, labelled_share):
    """ShardedGCN(labelled_logits_only='backward'): every logit is formed as the reference forms them, and the
    backward product of the logits runs over the columns of labelled nodes only (the other rows of the gradient
    are exactly zero).  Against the step that keeps every column: losses and weights equal to rounding (rows cut
    into chunks are cut elsewhere in the smaller matrix).  With 192 hidden units and few labels the zeros go one
    layer further: the hidden layer's gradient is formed, packed, for the neighbours of labelled nodes only, its
    weight gradient summed over those rows and its backward product run over their columns."""
    from gcn_amd import dist as gdist, sparse
    from gcn_amd.dist_train import ShardedGCN
    scale, f_in, classes = 13, 64, 8
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    x = sparse.synthetic_dense(n, f_in, seed=7)
    gen = torch.Generator().manual_seed(5)
    cls = torch.randint(0, classes, (n,), generator=gen)
    labels = torch.nn.functional.one_hot(cls, classes).to(torch.int32).cuda()
    mask = (torch.rand(n, generator=gen) < labelled_share).to(torch.int32).cuda()
    runs = []
    for mode in ('backward', False):
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
        model = ShardedGCN(shard, f_in, [hidden], classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.5, seed=9,
                           aggregate_first=False, receptive_field=False, labelled_logits_only=mode)
        losses = [float(model.train_step(x, labels, mask)) for _ in range(3)]
        logits = model.forward(x, train=False)
        assert bool((logits[mask == 0].abs().sum(dim=1) > 0).any())          # unlabelled nodes have logits too
        runs.append((losses, [w.detach().clone() for w in model.weights]))
        if mode == 'backward':
            backward_plan = next(iter(model._grad_plans.values()))[1][1]
            assert 0 < backward_plan.live_rows.numel() < n
            # where most nodes have no labelled neighbour the support restricted to the others' columns exists
            assert (backward_plan.live_support is not None) == (2 * backward_plan.live_rows.numel() < n)
            if hidden == 192:
                assert backward_plan.live_support is not None and backward_plan.live_support.nnz < shard.local.nnz
    assert np.allclose(runs[0][0], runs[1][0], rtol=2e-6)
    # Adam divides a gradient by its own running magnitude: where a weight's gradient is a sum that nearly cancels
    # (few labelled nodes), another order of its additions shows in the update as 1e-3 of the step size 0.02
    for a, b in zip(runs[0][1], runs[1][1]):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
        assert float((a - b).abs().mean()) <= 2e-6 * float(b.abs().max())
