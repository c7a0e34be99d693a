#!/usr/bin/env python3
"""Headline benchmark: C = A_hat . B on a synthetic R-MAT graph (the SpMM of gcn/layers.py:202).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the product over the whole graph with inputs resident in HBM.
Default workload: R-MAT scale 24 (2^24 nodes), edge factor 16 (average degree ~32 after
symmetrisation), 256-wide float32 features, A_hat = D^-1/2 (A+I) D^-1/2.  With N > 1 the rows
of A_hat, B and C are split in N contiguous blocks; every step each rank receives the rows of B its
block references (RCCL all-to-all, pipelined over ranges of feature columns) and multiplies locally
(strong scaling: the graph does not grow with N).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak, GB/s
EPOCH_LIMIT_S = 300           # the training-step leg gives up after this long (the product's line is printed regardless)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scale', type=int, default=24)
    ap.add_argument('--edge-factor', type=int, default=16)
    ap.add_argument('--features', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-epoch', action='store_true', help='skip the Planetoid epoch-time leg')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the other BASELINE configurations (scale 22 F=512, PubMed smoothing, narrow widths)')
    ap.add_argument('--chunks', type=int, default=4,
                    help='N>1: ranges of feature columns the exchange is pipelined over (1 = no overlap)')
    ap.add_argument('--no-autotune', action='store_true',
                    help='N>1: keep the --chunks progressive ranges instead of timing the candidate splits')
    ap.add_argument('--in-order', action='store_true',
                    help='N > 1: keep the single-GPU order of additions (bit-identical output) instead of multiplying '
                         'the entries that read own rows while the exchange runs')
    ap.add_argument('--allgather', action='store_true',
                    help='N > 1: all-gather the whole of B instead of exchanging only the referenced rows')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='development only: every rank uses cuda:0 and the gloo backend')
    ap.add_argument('--single-rank-collectives', action='store_true',
                    help='development only: one rank, but through the N-rank code path over RCCL (1-rank communicator)')
    ap.add_argument('--cpu-row-stride', type=int, default=16,
                    help='the CPU baseline multiplies every k-th row of the same matrix')
    return ap.parse_args()


def bytes_min(n_rows, n_cols, nnz, f):
    """Compulsory traffic of one product: (index + value) per nonzero, int64 row pointers,
    B read once, C written once."""
    return 8 * nnz + 8 * (n_rows + 1) + 4 * n_cols * f + 4 * n_rows * f


def bytes_gather(n_rows, nnz, f):
    """Upper traffic model: every nonzero fetches its own f-wide row of B."""
    return 8 * nnz + 8 * (n_rows + 1) + 4 * f * nnz + 4 * n_rows * f


def build_graph(scale, edge_factor):
    from gcn_amd import sparse
    indptr, indices = sparse.rmat_csr(scale, edge_factor, seed=42, self_loops=True)
    deg = sparse.row_sums(indptr, None)
    d = sparse.degree_scale(deg, 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    del deg, d
    torch.cuda.empty_cache()
    return indptr, indices, vals


def kernel_source_hash():
    """sha256 of the product kernels' source: a counter record is only quoted for the source it was taken on."""
    import hashlib
    with open(os.path.join(ROOT, 'gcn_amd', 'csrc', 'spmm.hip'), 'rb') as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def measured_traffic(kernel, scale, edge_factor, f, world):
    """Bytes per launch that the row kernel moves BEHIND THE L2 (read requests of the L2 to the fabric x
    128 bytes + WRITE_SIZE), from the rocprofv3 counter passes recorded in profiles/traffic.json.  This
    is L2-miss traffic: what the Infinity Cache serves is included, so it bounds the HBM bytes from
    above.  The record names the profile it comes from and the hash of the kernel source it was taken
    on; with another source, or no matching record, the answer is None."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        for rec in json.load(fh):
            if (rec['kernel'], rec['scale'], rec['edge_factor'], rec['features']) == (kernel, scale, edge_factor, f):
                # the record is quoted for the kernel source it was taken on, and only while the library that
                # is loaded was built from the sources on disk (the hash compiled into it says so)
                from gcn_amd import _lib
                if rec.get('spmm_hip_sha256') != kernel_source_hash() or _lib.build_hash() != _lib.source_hash():
                    return None
                return rec
    return None


def ranks_seen(sharded, rank, dev):
    """Who took part, as the communicator itself reports it: the world size of the process group, its
    backend, and the device every rank computes on (index, name and, where the runtime gives one, the
    device's uuid), gathered over the group -- so that a record of an N-GPU run can be checked for N
    distinct GPUs behind N RCCL ranks."""
    props = torch.cuda.get_device_properties(dev)
    mine = {'rank': rank, 'device_index': dev.index, 'name': props.name, 'uuid': str(getattr(props, 'uuid', None)),
            'pid': os.getpid()}
    if not sharded:
        return {'world_size': 1, 'backend': None, 'devices': [mine], 'distinct_devices': 1}
    import torch.distributed as dist
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    distinct = len({(d['device_index'], d['uuid']) for d in everyone})
    return {'world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'devices': everyone,
            'distinct_devices': distinct}


def host_description():
    """Core counts and CPU model of the host the CPU baseline runs on."""
    model = None
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None
    return {'os_cpu_count': os.cpu_count(), 'cpus_this_process_may_use': usable, 'cpu_model': model,
            'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS')}


def planetoid_epoch_ms(name, epochs=60, hip_graph=False):
    """Median wall time of one optimiser step (forward + backward + Adam: the region the reference
    times at train.py:203-205) of the 2-layer GCN on a Planetoid dataset, 20 labels per class."""
    import contextlib
    import io
    import train as driver
    from config import load_configuration
    from gcn_amd import graph as tf
    configuration, args = load_configuration(['--dataset', name, '--train_size', '20'] +
                                             (['--hip_graph'] if hip_graph else []), echo=False)
    driver.args = args
    cfg = configuration['model_list'][0]
    cfg['epochs'] = epochs
    np.random.seed(123)
    tf.set_random_seed(123)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), tf.Session() as sess:
        test_acc, _, _, _ = driver.train(cfg, sess, 123)
    ms = [float(line.split('median')[1].split()[0]) for line in buf.getvalue().splitlines() if 'epoch ms' in line]
    return {'dataset': name, 'epoch_ms': ms[0] if ms else None, 'epochs': epochs, 'test_acc': float(test_acc),
            'train_step': 'one HIP graph' if hip_graph else 'eager launches'}


def rmat_epoch_ms(shard, x_local, group_world, steps=10, warmup=2, hidden=256, classes=16, chunks=4, reorder=True,
                  own_first=False):
    """One training step (forward + backward + Adam: the region train.py:203-205 times) of a 2-layer GCN
    features -> hidden -> classes on the bench graph, rows sharded over the ranks, in two forms:

    ``reference_order_ms``: the reference's own order of operations -- every layer as A_hat . (X' . W)
    (gcn/layers.py:198-202) over ALL rows (logits of unlabelled nodes included, as models.py:54-69 computes
    them), with the backward products that order implies.  This is
    the "GCN epoch ms" of the metric.
    ``ms``: the same step, same loss and weights to rounding, restructured: first layer as (A_hat . X') . W0
    and hidden rows formed only where a labelled node reads them."""
    import torch
    from gcn_amd.dist_train import ShardedGCN
    r0, r1 = shard.row_begin, shard.row_end
    rows = torch.arange(r0, r1, device=x_local.device)
    labels = torch.nn.functional.one_hot((rows * 2654435761 >> 7) % classes, classes).to(torch.int32)
    mask = (rows % 64 == 0).to(torch.int32)
    del rows

    def timed(**form):
        model = ShardedGCN(shard, x_local.shape[1], [hidden], classes, learning_rate=0.02, weight_decay=5e-4,
                           dropout=0.5, seed=123, chunks=chunks, own_first=own_first, **form)
        for _ in range(warmup):
            model.train_step(x_local, labels, mask)
        per_step = []
        for _ in range(steps):                      # every step bracketed like the whole run: barrier + synchronize
            if group_world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = model.train_step(x_local, labels, mask)
            torch.cuda.synchronize()
            per_step.append((time.perf_counter() - t0) * 1e3)
        t = torch.tensor(per_step, dtype=torch.float64, device=x_local.device)
        if group_world > 1:                          # a step takes as long as its slowest rank
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        per_step = t.cpu().numpy()
        described = (model.aggregate_first, model.receptive_field and model.aggregate_first)
        loss = float(loss)
        del model
        torch.cuda.empty_cache()
        return float(np.median(per_step)), float(per_step.min()), float(per_step.mean()), loss, described

    # reference order: every product of the forward pass as the reference makes it, over all rows; the backward
    # products skip what is exactly zero in their operands (the hidden layer's gradient is gathered packed, the
    # gradient of the logits is non-zero at labelled nodes only)
    ref_ms, ref_min, ref_mean, ref_loss, _ = timed(aggregate_first=False, receptive_field=False,
                                                   labelled_logits_only='backward')
    ms, ms_min, ms_mean, loss, (aggregated, pruned) = timed()
    dense = None
    if group_world == 1:
        # the same two steps with every product on dense operands (packed_operands=False), for the comparison
        d_ref = timed(aggregate_first=False, receptive_field=False, labelled_logits_only=False, packed_operands=False)
        d_new = timed(packed_operands=False)
        dense = {'reference_order_ms': d_ref[0], 'reference_order_loss': d_ref[3], 'ms': d_new[0], 'loss': d_new[3]}
    reordered = None
    if group_world == 1 and reorder:
        # the same two steps on the graph relabelled hub-first (an integer preprocessing step of this build;
        # inputs permuted once, outside the timed region)
        r_ref = timed(aggregate_first=False, receptive_field=False, labelled_logits_only='backward', reorder='hubs')
        r_new = timed(reorder='hubs')
        reordered = {'vertex_order': "hub-first: the 65,536 vertices of highest degree first, the others in their old order",
                     'reference_order_ms': r_ref[0], 'reference_order_ms_min': r_ref[1], 'reference_order_loss': r_ref[3],
                     'ms': r_new[0], 'ms_min': r_new[1], 'loss': r_new[3]}
    return {'hub_first_order': reordered,
            'reference_order_ms': ref_ms, 'reference_order_ms_min': ref_min, 'reference_order_ms_mean': ref_mean,
            'reference_order_loss': ref_loss, 'ms': ms, 'ms_min': ms_min, 'ms_mean': ms_mean, 'loss': loss,
            'statistic': 'median of %d timed steps (min and mean beside it), each between synchronisations' % steps,
            'layers': [int(x_local.shape[1]), hidden, classes], 'steps': steps,
            'what': 'forward + backward + Adam, dropout 0.5, one labelled node in 64, rows/%d.  reference_order_ms: '
                    "every layer as A.(X'.W) over all rows (gcn/layers.py:198-202).  ms: first layer as %s%s" % (
                        group_world, "(A.X').W0 (no backward product)" if aggregated else "A.(X'.W0)",
                        ', hidden rows formed only where a labelled node reads them' if pruned else '')
                    + ('.  One GPU: the operands that are mostly zeros -- dropout(X) in ms, the hidden layer\'s gradient '
                       'in reference_order_ms -- are gathered in packed form, and the backward pass of reference_order_ms '
                       'skips what is exactly zero: the backward product of the logits runs over the columns of labelled '
                       'nodes, the hidden layer\'s gradient is formed, and its products run, for the neighbours of labelled '
                       'nodes only (every logit is still formed forward; same values, sums to rounding); dense_operands: '
                       'the same steps with every operand dense and every row and column kept'
                       if group_world == 1 else
                       '.  N ranks: in ms the rows of X that a shard reads from other ranks are fetched once, before the '
                       'timed steps (X is an input that does not change between steps, like the support), and every step '
                       'drops own and fetched rows locally by their global row numbers into one packed operand: the first '
                       'layer exchanges nothing; the narrow products of the classes layer exchange their rows every step.  '
                       'reference_order_ms exchanges the wide rows of X\'.W0 every step, and of the hidden layer\'s gradient the rows of '
                       'labelled nodes\' neighbours (the others are zero)'),
            'dense_operands': dense}


def cpu_baseline(a, b_dev, c_dev, row_stride):
    """Time the CPU oracle (the serial nnz loop of the reference's TensorFlow kernel, restated in C)
    on every `row_stride`-th row of the same matrix against the same B, and check the GPU rows."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gcn_oracle as oracle
    n = a.n_rows
    rows = torch.arange(0, n, row_stride, device=a.device)
    starts = a.indptr[rows]
    lens = a.indptr[rows + 1] - starts
    sub_indptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=a.device)
    sub_indptr[1:] = torch.cumsum(lens, 0)
    nnz_s = int(sub_indptr[-1].item())
    pos = torch.repeat_interleave(starts - sub_indptr[:-1], lens) + torch.arange(nnz_s, device=a.device)
    ip = sub_indptr.cpu().numpy()
    ix = a.indices[pos].cpu().numpy()
    vv = a.values[pos].cpu().numpy()
    del pos
    b = b_dev.cpu().numpy()
    f = b.shape[1]
    out = np.empty((rows.numel(), f), dtype=np.float32)
    t0 = time.perf_counter()
    oracle.spmm_csr(ip, ix, vv, b, threads=1, out=out)
    t_serial = time.perf_counter() - t0
    # every thread OpenMP gives this process (the box caps OMP_NUM_THREADS at its CPU share)
    threads = oracle._load_clib().oracle_num_threads()
    t0 = time.perf_counter()
    out_mt = oracle.spmm_csr(ip, ix, vv, b, threads=threads)
    t_mt = time.perf_counter() - t0
    gpu_rows = c_dev[rows].cpu().numpy()
    short = np.diff(ip) <= a.plan.split_nnz
    exact = bool(np.array_equal(gpu_rows[short], out[short]) and np.array_equal(out, out_mt))
    scale_ref = float(np.abs(out).max())
    max_err = float(np.abs(gpu_rows - out).max() / scale_ref)
    full = bytes_min(n, a.n_cols, a.nnz, f)
    frac = nnz_s / a.nnz
    sample = ('every %d-th row of the same A_hat (%d rows, %d nonzeros = %.4f of the matrix) times the '
              'same B; GB/s = that fraction of the full product\'s algorithmic bytes / time'
              % (row_stride, rows.numel(), nnz_s, frac))
    base = {'value': full * frac / t_serial / 1e9, 'unit': 'GB/s', 'cores': 1, 'kind': 'port',
            'sample': sample, 'seconds': t_serial}
    allcores = {'value': full * frac / t_mt / 1e9, 'unit': 'GB/s', 'cores': threads, 'kind': 'port',
                'sample': sample, 'seconds': t_mt}
    host = host_description()
    base['host'] = host
    allcores['host'] = host
    check = {'rows_checked': int(rows.numel()), 'short_rows_bit_exact': exact, 'max_rel_err': max_err}
    return base, allcores, check


def _median_min(ms):
    return float(np.median(ms)), float(np.min(ms))


def product_point(label, a, f, steps=10, warmup=2, cpu_rows=0):
    """One more point of the product: C = a . B for a seeded B of f columns, resident in HBM.  Per launch
    the row kernel's own duration (HIP events recorded by the library on the launch stream) and the roofline
    triple on the algorithmic bytes."""
    from gcn_amd import sparse
    b = sparse.synthetic_dense(a.n_cols, f, seed=7)
    c = torch.empty((a.n_rows, f), dtype=torch.float32, device=b.device)
    a.plan.set_timing(True)
    for _ in range(warmup):
        sparse.spmm(a, b, out=c)
    torch.cuda.synchronize()
    wall, kern = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        sparse.spmm(a, b, out=c)
        row_ms, red_ms = a.plan.last_timing()          # waits for the product
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(row_ms)
    a.plan.set_timing(False)
    del b, c
    torch.cuda.empty_cache()
    algo = bytes_min(a.n_rows, a.n_cols, a.nnz, f)
    gathered = bytes_gather(a.n_rows, a.nnz, f)
    k_med, k_min = _median_min(kern)
    w_med, w_min = _median_min(wall)
    achieved = algo / (k_med * 1e-3) / 1e9
    return {'workload': label, 'features': f, 'n_nodes': a.n_rows, 'nnz': a.nnz, 'steps': steps,
            'ms_per_product': w_med, 'ms_min': w_min, 'value_gbps': algo / (w_med * 1e-3) / 1e9,
            'gather_gbps': gathered / (w_med * 1e-3) / 1e9,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': None, 'kernel': 'spmm_rows_kernel',
                         'kernel_ms': k_med, 'kernel_ms_min': k_min, 'algorithmic_bytes': algo}}


def packed_point(a, scale, edge_factor, f, keep, steps=10, warmup=2):
    """C = a . dropout(X, keep) with dropout(X) written and gathered in packed form (its kept values only): the
    product of the first layer of a training step.  Algorithmic bytes: the CSR, the packed operand once
    (64 + 4 kept bytes per row), C once."""
    from gcn_amd import sparse
    x = sparse.synthetic_dense(a.n_cols, f, seed=7)
    packed = sparse.PackedRows.empty(a.n_cols, f, x.device)
    c = torch.empty((a.n_rows, f), dtype=torch.float32, device=x.device)
    pack_ms = []
    for i in range(warmup + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sparse.dropout_pack(x, keep, 5, 1, out=packed)
        torch.cuda.synchronize()
        if i >= warmup:
            pack_ms.append((time.perf_counter() - t0) * 1e3)
    kept = int(packed.kept().sum().item())
    del x
    a.plan.set_timing(True)
    for _ in range(warmup):
        sparse.spmm_packed(a, packed, out=c)
    torch.cuda.synchronize()
    kern = []
    for _ in range(steps):
        sparse.spmm_packed(a, packed, out=c)
        kern.append(a.plan.last_timing()[0])
    a.plan.set_timing(False)
    del packed, c
    torch.cuda.empty_cache()
    k_med, k_min = _median_min(kern)
    algo = 8 * a.nnz + 8 * (a.n_rows + 1) + 64 * a.n_cols + 4 * kept + 4 * a.n_rows * f
    achieved = algo / (k_med * 1e-3) / 1e9
    return {'workload': 'A_hat . dropout(X, keep %.2f), R-MAT scale %d edge factor %d, F=%d, dropout(X) packed (kept '
                        'values only)' % (keep, scale, edge_factor, f), 'features': f, 'n_nodes': a.n_rows, 'nnz': a.nnz,
            'kept_values': kept, 'steps': steps, 'ms_per_product': k_med, 'ms_min': k_min,
            'pack_ms': float(np.median(pack_ms)),
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': None, 'kernel': 'spmm_packed_kernel',
                         'kernel_ms': k_med, 'kernel_ms_min': k_min, 'algorithmic_bytes': algo}}


def pubmed_support(device):
    """A_hat of the PubMed graph from the committed vectors of the reference's own preprocess_adj."""
    from gcn_amd import sparse
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'pubmed_graph.npz'))
    n = int(fx['n'])
    ip, ix = fx['ai_indptr'].astype(np.int64), fx['ai_indices'].astype(np.int32)
    v = fx['hat_values'].astype(np.float32)
    a = sparse.DeviceCSR(n, n, torch.as_tensor(ip).to(device), torch.as_tensor(ix).to(device),
                         torch.as_tensor(v).to(device))
    return a, (ip, ix, v)


def smoothing_point(device, k=64, f=500, replays=20, with_cpu=True):
    """A_hat^k X on PubMed (X seeded, 19,717 x 500), the k products recorded as ONE HIP graph and replayed:
    microseconds per step = replay time / k.  Beside it the same step of the CPU oracle on one core."""
    from gcn_amd import smoothing, sparse
    a, (ip, ix, v) = pubmed_support(device)
    x = sparse.synthetic_dense(a.n_rows, f, seed=7)
    run = smoothing.GraphedPropagation(a, f, k)
    for _ in range(3):
        run(x)
    torch.cuda.synchronize()
    us = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run.graph.replay()
        e1.record()
        e1.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / k)
    med, low = _median_min(us)
    algo = bytes_min(a.n_rows, a.n_cols, a.nnz, f)
    achieved = algo / (med * 1e-6) / 1e9
    out = {'workload': 'A_hat^k X on PubMed (A_hat from the reference vectors, X synthetic 19717 x %d), k = %d as one HIP graph'
                       % (f, k), 'k': k, 'features': f, 'n_nodes': a.n_rows, 'nnz': a.nnz, 'replays': replays,
           'us_per_step': med, 'us_per_step_min': low,
           'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                        'frac': achieved / HBM_PEAK_GBPS, 'traffic': None, 'kernel': 'spmm_rows_kernel',
                        'kernel_ms': med * 1e-3, 'algorithmic_bytes': algo,
                        'note': 'operand and result (79 MB) stay in the 256 MiB Infinity Cache between steps: '
                                'this measures launch-to-launch latency and the gather path, not HBM'}}
    # the same single step with operands from HBM: a 512 MB memset (twice the Infinity Cache) before every
    # repetition, events around the product alone
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=device)
    c = torch.empty_like(x)
    cold = []
    for _ in range(replays):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sparse.spmm(a, x, out=c)
        e1.record()
        e1.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    out['us_per_step_cold'], out['us_per_step_cold_min'] = _median_min(cold)
    out['cold'] = ('one step launched on its own after a 512 MB memset: operand, matrix and result come from HBM and '
                   'the time includes one launch')
    del flush, c
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import gcn_oracle as oracle
        xh = x.cpu().numpy()
        out_h = np.empty_like(xh)
        oracle.spmm_csr(ip, ix, v, xh, threads=1, out=out_h)          # untimed first pass
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            oracle.spmm_csr(ip, ix, v, xh, threads=1, out=out_h)
            t.append(time.perf_counter() - t0)
        sec = float(np.median(t))
        out['cpu_baseline'] = {'value': sec * 1e6, 'unit': 'us per step', 'cores': 1, 'kind': 'port',
                               'sample': 'one step A_hat . X of the same size (median of 5), the serial loop of the '
                                         'oracle', 'gbps': algo / sec / 1e9}
    del run, x
    torch.cuda.empty_cache()
    return out


def cpu_planetoid_step(name, steps=8):
    """One optimiser step (forward + backward + Adam, dropout 0.5: the region train.py:203-205 times) of
    the CPU oracle -- the reference's arithmetic restated in numpy / C, one core -- on a Planetoid dataset
    with the 20-labels-per-class split of the committed fixture."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gcn_oracle as oracle
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    n = int(fx['n'])
    ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    feat = (fx['feat_indptr'].astype(np.int64), fx['feat_coords'][:, 1], fx['feat_values'].astype(np.float32))
    f_in, classes = int(fx['feat_shape'][1]), int(fx['y_train'].shape[1])
    rng = np.random.RandomState(123)
    weights = [oracle.glorot((f_in, 16), rng), oracle.glorot((16, classes), rng)]
    model = oracle.GcnOracle((ip, ix, v32), feat, n, weights)
    adam = oracle.TfAdam(model.w, 0.02)
    labels, mask = fx['y_train'], fx['train_mask']
    ms = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        logits = model.forward(dropout=0.5, seed=123, stream_ids=[2 * step, 2 * step + 1])
        grads = model.backward(logits, labels, mask)
        adam.step(model.w, grads)
        if step:                                   # the first step is untimed
            ms.append((time.perf_counter() - t0) * 1e3)
    med, low = _median_min(ms)
    return {'dataset': name, 'value': med, 'min': low, 'unit': 'ms per step', 'cores': 1, 'kind': 'port',
            'sample': '%d optimiser steps of the oracle (numpy + the C loop of the sparse products), '
                      'one thread; first step untimed' % steps}


def pubmed_step(with_cpu=True, steps=40):
    """One optimiser step (forward + backward + Adam, dropout 0.5: the region train.py:203-205 times) of the 2-layer
    GCN on PubMed -- A_hat from the committed vectors of the reference's own preprocess_adj, 60 labelled nodes,
    X a seeded synthetic 19,717 x 500 matrix with ~50 entries per row fed as a sparse triple (the dataset's
    feature file is absent upstream) --, launched kernel by kernel and as one HIP graph; beside it the same step
    of the CPU oracle on one core."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gcn_oracle as oracle
    from gcn_amd import graph as tf
    from gcn_amd.models import GCN_MLP
    from gcn_amd.utils import SupportValue, construct_feed_dict
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'pubmed_graph.npz'))
    n, f_in, classes = int(fx['n']), 500, 3
    x = oracle.synthetic_dense(n, f_in, seed=19)
    x = np.where(x > 0.8, x, np.float32(0))
    r, c = np.nonzero(x)
    features = SupportValue(np.stack([r, c], 1).astype(np.int32), x[r, c], np.array([n, f_in], dtype=np.int64))
    support = [SupportValue(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])]
    rng = np.random.RandomState(8)
    labels = np.eye(classes)[rng.randint(0, classes, n)]
    mask = np.zeros(n, dtype=bool)
    mask[rng.choice(n, 60, replace=False)] = True
    out = {'dataset': 'pubmed (graph from the reference vectors, synthetic features: %d entries)' % r.size,
           'labelled': 60, 'steps': steps}
    for recorded in (False, True):
        tf.set_random_seed(41)
        placeholders = {
            'support': [tf.sparse_placeholder(tf.float32, name='support0')],
            'features': tf.sparse_placeholder(tf.float32, name='features'),
            'labels': tf.placeholder(tf.int32, name='labels', shape=(None, classes)),
            'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
            'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
            'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
        }
        cfg = {'name': 'gcn', 'connection': ['c', 'c'], 'layer_size': [16], 'conv': 'gcn', 'learning_rate': 0.02,
               'weight_decay': 5e-4, 'Model': 'GCN', 'dropout': 0.5, 'hip_graph': recorded}
        model = GCN_MLP(cfg, placeholders, input_dim=f_in)
        feed = construct_feed_dict(features, support, labels, mask, placeholders)
        feed[placeholders['dropout']] = 0.5
        sess = tf.Session()
        ms = []
        for step in range(steps + 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sess.run(model.opt_op, feed_dict=feed)
            torch.cuda.synchronize()
            if step >= 5:
                ms.append((time.perf_counter() - t0) * 1e3)
        loss = float(sess.run(model.loss, feed_dict=feed))
        key = 'one_hip_graph' if recorded else 'eager_launches'
        out[key] = {'epoch_ms': float(np.median(ms)), 'epoch_ms_min': float(np.min(ms)), 'loss_after': loss}
        del model, sess
    if with_cpu:
        ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
        counts = np.bincount(r, minlength=n)
        feat = (np.concatenate([[0], np.cumsum(counts)]).astype(np.int64), c.astype(np.int32), x[r, c].astype(np.float32))
        rs = np.random.RandomState(123)
        weights = [oracle.glorot((f_in, 16), rs), oracle.glorot((16, classes), rs)]
        model = oracle.GcnOracle((ip, ix, v32), feat, n, weights)
        adam = oracle.TfAdam(model.w, 0.02)
        ms = []
        for step in range(7):
            t0 = time.perf_counter()
            logits = model.forward(dropout=0.5, seed=123, stream_ids=[2 * step, 2 * step + 1])
            grads = model.backward(logits, labels, mask)
            adam.step(model.w, grads)
            if step:
                ms.append((time.perf_counter() - t0) * 1e3)
        out['cpu_baseline'] = {'value': float(np.median(ms)), 'min': float(np.min(ms)), 'unit': 'ms per step', 'cores': 1,
                               'kind': 'port', 'sample': '6 optimiser steps of the oracle (numpy + the C loop of the '
                                                         'sparse products), one thread; first step untimed'}
    return out


def other_configs(a24, scale, edge_factor, device, with_cpu):
    """The other BASELINE configurations and the narrower widths of the headline graph, timed in this run."""
    from gcn_amd import sparse
    out = []
    for f in (16, 64, 128):
        out.append(product_point('A_hat.B, R-MAT scale %d edge factor %d, F=%d (the widths a GCN layer '
                                 'propagates)' % (scale, edge_factor, f), a24, f))
    # the same graph relabelled hub-first (gcn_hub_order + gcn_csr_relabel: an integer preprocessing step)
    from gcn_amd.utils import relabelled_csr, vertex_order
    hubs = relabelled_csr(a24, vertex_order(a24.indptr, 'hubs')[0])
    torch.cuda.empty_cache()
    for f in (16, 64, 256):
        out.append(product_point('A_hat.B, R-MAT scale %d edge factor %d relabelled hub-first (65,536 hubs), F=%d'
                                 % (scale, edge_factor, f), hubs, f))
    del hubs
    torch.cuda.empty_cache()
    out.append(packed_point(a24, scale, edge_factor, 256, 0.5))
    # BASELINE config 4: R-MAT 2^22, average degree 32, 512 features
    ip, ix, v = build_graph(22, 16)
    a22 = sparse.DeviceCSR(1 << 22, 1 << 22, ip, ix, v)
    out.append(product_point('BASELINE config 4: A_hat.B, R-MAT scale 22 edge factor 16, F=512', a22, 512))
    del a22, ip, ix, v
    torch.cuda.empty_cache()
    # BASELINE config 3: k-step smoothing on PubMed
    out.append(smoothing_point(device, with_cpu=with_cpu))
    return out


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d: launch with torch.distributed.run'
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sharded = world > 1 or args.single_rank_collectives
    if sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29433')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        elif args.rehearse_on_one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from gcn_amd import _lib, sparse, dist as gdist
    seen = ranks_seen(sharded, rank, dev)

    n = 1 << args.scale
    f = args.features

    if not sharded:
        indptr, indices, vals = build_graph(args.scale, args.edge_factor)
        nnz = int(indices.numel())
        b_full = sparse.synthetic_dense(n, f, seed=7)
        a = sparse.DeviceCSR(n, n, indptr, indices, vals)
        c = torch.empty((n, f), dtype=torch.float32, device=dev)
        a.plan.set_timing(True)
        shard = gdist.RowShardedCSR(n, n, 0, n, a, 0, 1)
        b_local = b_full

        def step():
            sparse.spmm(a, b_full, out=c)
    else:
        # every rank generates its own row block of A_hat and of B and nothing else
        shard = gdist.rmat_shard(args.scale, args.edge_factor, rank, world, seed=42)
        b_local = sparse.synthetic_dense(shard.row_end - shard.row_begin, f, seed=7, first_row=shard.row_begin)
        t = torch.tensor([shard.local.nnz], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t)
        nnz = int(t.item())
        torch.cuda.empty_cache()
        own_first = False if args.in_order else 'auto'
        tuned = None
        if args.allgather:
            op = gdist.AllGatherSpmm(shard, f)
        elif args.no_autotune or args.in_order or world == 1:
            op = gdist.HaloSpmm(shard, f, chunks=args.chunks, own_first=own_first)
            if not args.no_autotune:
                tuned = [(r, t, op.plan.own_first) for r, t in op.autotune(b_local)]
        else:
            # Which form pipelines best depends on what the links of THIS node deliver beside the product: both
            # orders of additions (single-GPU order; own rows first, which hides the exchange behind work that
            # needs none) and every candidate split of the feature columns are timed once on this communicator --
            # maximum over the ranks, so every rank takes the same decision -- and the fastest is kept.
            tuned, best = [], None
            for first in (False, True):
                cand = gdist.HaloSpmm(shard, f, chunks=args.chunks, own_first=first)
                timings = cand.autotune(b_local)
                tuned += [(r, t, first) for r, t in timings]
                fastest = min(t for _, t in timings)
                if best is None or fastest < best[0]:
                    if best is not None:
                        best[1].plan.drop_buffers()
                    best = (fastest, cand)
                else:
                    cand.plan.drop_buffers()
                del cand
                torch.cuda.empty_cache()
            op = best[1]
            del best
        # the matrix whose kernel the roofline entry times: the one the step's last launch multiplies
        a = op.plan.matrices[-1] if not args.allgather else shard.local
        a.plan.set_timing(True)

        def step():
            op(b_local)

    def barrier():
        if sharded:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(a.plan.last_timing())          # waits for this step's kernels (per-step event read)
    barrier()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        total_min = bytes_min(n, n, nnz, f)
        total_gather = bytes_gather(n, nnz, f)
        value = total_min / (ms_per_step * 1e-3) / 1e9
        row_ms = float(np.mean([k[0] for k in kernel_ms]))
        red_ms = float(np.mean([k[1] for k in kernel_ms]))
        # the row kernel of THIS rank: its shard of the rows against the whole of B
        # (pipelined over column ranges, the events bracket the launch of the last range)
        f_launch = f
        if sharded and not args.allgather:
            lo, hi = op.ranges[-1]
            f_launch = hi - lo
        local_min = bytes_min(a.n_rows, a.n_cols, a.nnz, f_launch)
        achieved = local_min / (row_ms * 1e-3) / 1e9
        line = {
            'metric': 'SpMM HBM GB/s', 'value': value, 'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'A_hat.B, R-MAT scale %d edge factor %d (+I, sym-normalised), F=%d'
                                   % (args.scale, args.edge_factor, f),
                       'n_nodes': n, 'nnz': nnz, 'features': f, 'ranks_seen': seen,
                       'library_build_hash': _lib.build_hash(),
                       'library_matches_sources': _lib.build_hash() == _lib.source_hash(),
                       'parallelism': ('rows/%d + %s' % (world, 'all-gather(B)' if args.allgather else 'exchange of referenced rows of B pipelined over column ranges %s%s' % (op.ranges, ', own rows multiplied first' if op.plan.own_first else ''))) if sharded else 'single GPU'},
            'pct_of_hbm_peak': 100.0 * value / (HBM_PEAK_GBPS * world),
            'gather_gbps': total_gather / (ms_per_step * 1e-3) / 1e9,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': None,
                         'kernel': 'spmm_rows_kernel', 'kernel_ms': row_ms, 'reduce_kernel_ms': red_ms,
                         'algorithmic_bytes': local_min, 'launch_features': f_launch},
            'pipeline_candidates_ms': ([{'ranges': r, 'ms': t, 'own_rows_first': o} for r, t, o in tuned]
                                       if (sharded and tuned) else None),
        }
        if not sharded and not args.no_cpu_baseline:
            base, allcores, check = cpu_baseline(a, b_full, c, args.cpu_row_stride)
            line['cpu_baseline'] = base
            line['cpu_baseline_all_cores'] = allcores
            line['parity_sample'] = check
        if not sharded and not args.no_epoch:
            line['gcn_epoch'] = []
            for name, recorded in (('cora', False), ('citeseer', False), ('cora', True), ('citeseer', True)):
                try:
                    line['gcn_epoch'].append(planetoid_epoch_ms(name, hip_graph=recorded))
                except Exception as exc:
                    line['gcn_epoch'].append({'dataset': name, 'error': '%s: %s' % (type(exc).__name__, exc)})
            try:                                  # BASELINE config 2's other dataset
                line['gcn_epoch'].append(pubmed_step(with_cpu=not args.no_cpu_baseline))
            except Exception as exc:
                line['gcn_epoch'].append({'dataset': 'pubmed', 'error': '%s: %s' % (type(exc).__name__, exc)})
            if not args.no_cpu_baseline:
                # the same step of the CPU oracle on one core, beside the GPU's
                line['gcn_epoch_cpu_baseline'] = []
                for name in ('cora', 'citeseer'):
                    try:
                        line['gcn_epoch_cpu_baseline'].append(cpu_planetoid_step(name))
                    except Exception as exc:
                        line['gcn_epoch_cpu_baseline'].append({'dataset': name, 'error': '%s: %s' % (type(exc).__name__, exc)})
        if not sharded and not args.no_configs:
            try:
                a.plan.set_timing(False)
                line['configs'] = other_configs(a, args.scale, args.edge_factor, dev, not args.no_cpu_baseline)
            except Exception as exc:
                line['configs'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        rec = measured_traffic('spmm_rows_kernel', args.scale, args.edge_factor, f, world)
        if rec:
            # bytes behind the L2 per launch (Infinity-Cache hits included: an upper bound of the HBM bytes)
            line['roofline']['traffic'] = rec['l2_miss_bytes_per_launch']
            line['roofline']['traffic_source'] = rec['source']
            line['l2_miss_traffic_gbps'] = rec['l2_miss_bytes_per_launch'] / (row_ms * 1e-3) / 1e9
    ranges_used = list(op.ranges) if (sharded and not args.allgather) else None
    # the training step takes the order of additions the product's measurement picked
    epoch_own_first = bool(op.plan.own_first) if (sharded and not args.allgather) else False
    # GCN epoch on the same graph, every rank takes part
    epoch = None
    printed = threading.Lock()
    watchdog = None
    if not args.no_epoch:
        # A rank that fails alone inside the training step leaves the others waiting in a collective:
        # the product's line must still come out, so a timer thread prints it and ends the process.
        def give_up():
            if printed.acquire(False):
                if rank == 0:
                    line['gcn_epoch_rmat'] = {'error': 'no result within %d s' % EPOCH_LIMIT_S}
                    print(json.dumps(line), flush=True)
                os._exit(3)                 # the product's line is out; the hang itself is a failure
        watchdog = threading.Timer(EPOCH_LIMIT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        a.plan.set_timing(False)
        if not sharded:
            del c
        else:
            del op
        torch.cuda.empty_cache()
        try:
            shares = args.chunks
            if sharded and not args.allgather:
                shares = tuple(hi - lo for lo, hi in ranges_used)
            epoch = rmat_epoch_ms(shard, b_local, world, chunks=shares, own_first=epoch_own_first)
        except Exception as exc:            # the product's line above stays valid whatever happens here
            epoch = {'error': '%s: %s' % (type(exc).__name__, exc)}
    if watchdog is not None:
        watchdog.cancel()
    if not printed.acquire(False):
        time.sleep(60)                      # the timer thread is printing and ends the process
    failed = epoch is not None and 'error' in epoch
    if rank == 0:
        if epoch is not None:
            line['gcn_epoch_rmat'] = epoch
        failed = failed or any('error' in leg for leg in line.get('gcn_epoch', []))
        failed = failed or any('error' in leg for leg in line.get('gcn_epoch_cpu_baseline', []))
        failed = failed or 'error' in line.get('configs', [])
        print(json.dumps(line), flush=True)
    if sharded:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if failed:
        sys.exit(1)                         # a leg raised: the line above says which; do not report success


if __name__ == '__main__':
    main()
