"""One rank's share of the row-sharded product, measured on one GPU: rank 0's shard of the scale-24 R-MAT
graph for 2, 4 and 8 ranks -- rows of B it references, local product at full width and in pipelined
column ranges, packing of a send buffer.  (No collectives: xGMI time is not part of this.)"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse, dist as gdist
scale, f = 24, 256
n = 1 << scale
indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ms=[]
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))
for world in (2, 4, 8):
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, world)
    local = shard.local
    cols = local.indices.to(torch.int64)
    needed = torch.unique(cols)
    new_cols = torch.searchsorted(needed, cols).to(torch.int32)
    compact = sparse.DeviceCSR(local.n_rows, int(needed.numel()), local.indptr, new_cols, local.values)
    per = n // world
    own = int(((needed >= 0) & (needed < per)).sum())
    n_needed = int(needed.numel())
    del cols, new_cols
    halo = sparse.synthetic_dense(n_needed, f, seed=3)
    out = torch.empty((local.n_rows, f), device='cuda')
    t_full = timed(lambda: sparse.spmm(compact, halo, out=out))
    halo64 = [torch.empty((n_needed, 64), device='cuda').copy_(halo[:, i*64:(i+1)*64]) for i in range(4)]
    def chunks():
        for i in range(4):
            sparse.spmm(compact, halo64[i], out=out[:, i*64:(i+1)*64], wide_split=True)
    t_chunks = timed(chunks)
    rng = gdist.column_chunks(f, 4)
    halop = [torch.empty((n_needed, hi-lo), device='cuda').copy_(halo[:, lo:hi]) for lo, hi in rng]
    def prog():
        for (lo, hi), hb in zip(rng, halop):
            sparse.spmm(compact, hb, out=out[:, lo:hi], wide_split=True)
    t_prog = timed(prog)
    print('world', world, 'progressive ranges', rng, '%.2f ms' % t_prog, flush=True)
    # send side: this rank packs about as many rows as it receives from others
    n_send = n_needed - own
    b_local = sparse.synthetic_dense(per, f, seed=5)
    rows = torch.randint(0, per, (n_send,), device='cuda')
    send = torch.empty((n_send, 64), device='cuda')
    t_pack = timed(lambda: sparse.gather_rows(b_local[:, 0:64], rows, send))
    print('world %d: rows/rank %d nnz %d needed %d (%.1f%% of N; own %d) | full-width product %.2f ms | 4 x 64 columns %.2f ms | pack of one 64-col range (%d rows) %.3f ms | received per range %.2f GB'
          % (world, local.n_rows, local.nnz, n_needed, 100.0*n_needed/n, own, t_full, t_chunks, n_send, t_pack, (n_needed-own)*256/1e9), flush=True)
    del shard, local, compact, halo, out, halo64, b_local, rows, send, needed
    torch.cuda.empty_cache()
