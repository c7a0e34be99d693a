"""One rank's share of the FIRST LAYER of the restructured training step, measured on ONE GPU: S = A[R, :] . dropout(X)
for rank 0 of 2, 4 and 8 ranks of the scale-24 R-MAT graph (R: the rows of its block that a labelled node reads,
one labelled node in 64), 256 features, in the two forms a rank can run it:

  kept:  the rows of X that the shard reads from other ranks are resident (fetched once, X is constant); own and kept
         rows are dropped by their global row numbers into ONE packed operand and the product gathers packed rows.
         No exchange in the step.
  sent:  dropout(X) of the own rows as a dense pass, the referenced rows packed for the exchange, a device copy
         standing in for the links (memory traffic of send + receive, not link time), the dense two-part product.

Printed per rank count: rows and entries of the rank's share, the two forms' times, the bytes the `sent` form moves
over the links per step (and the time they take at the nominal 153 GB/s per link with P - 1 links), and the one-GPU
step's first-layer time / P for comparison.

    python tools/emulate_step.py [--scale 24] [--out gpurun_out/r3/step_emulation.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse, dist as gdist   # noqa: E402

LINK_GBPS = 153.0


def timed(fn, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=24)
ap.add_argument('--out', default=None)
args = ap.parse_args()
n, f, keep = 1 << args.scale, 256, 0.5
dev = torch.device('cuda')
indptr, indices = sparse.rmat_csr(args.scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
del d
# the rows some labelled node reads (every rank's labelled rows count: the set is global)
labelled = torch.arange(0, n, 64, device=dev)
starts, lens = indptr[labelled], indptr[labelled + 1] - indptr[labelled]
base = torch.zeros(labelled.numel() + 1, dtype=torch.int64, device=dev)
base[1:] = torch.cumsum(lens, 0)
pos = torch.repeat_interleave(starts - base[:-1], lens) + torch.arange(int(base[-1]), device=dev)
seen = torch.zeros(n, dtype=torch.bool, device=dev)
seen[indices[pos].to(torch.int64)] = True
del pos, base, starts, lens

# the one-GPU step's first layer: pack + packed product over the rows in R
whole = gdist.RowShardedCSR(n, n, 0, n, sparse.DeviceCSR(n, n, indptr, indices, vals), 0, 1)
r_all = torch.nonzero(seen).reshape(-1)
wide1 = gdist.shard_rows_compacted(whole, r_all).local
x = sparse.synthetic_dense(n, f, seed=7)
packed = sparse.PackedRows.empty(n, f, dev)
t_pack1 = timed(lambda: sparse.dropout_pack(x, keep, 5, 1, out=packed))
out1 = torch.empty((wide1.n_rows, f), device=dev)
t_prod1 = timed(lambda: sparse.spmm_packed(wide1, packed, out=out1))
print('one GPU: %d of %d rows read by a labelled node, %d entries; pack %.2f ms + packed product %.2f ms = %.2f ms'
      % (wide1.n_rows, n, wide1.nnz, t_pack1, t_prod1, t_pack1 + t_prod1), flush=True)
results = {'one_gpu': {'rows': wide1.n_rows, 'nnz': wide1.nnz, 'pack_ms': t_pack1, 'product_ms': t_prod1}}
del wide1, out1, packed, x
torch.cuda.empty_cache()

for world in (2, 4, 8):
    per = n // world
    shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, world)
    r_local = torch.nonzero(seen[:per]).reshape(-1)
    local = gdist.shard_rows_compacted(shard, r_local).local
    cols = local.indices.to(torch.int64)
    mine = cols < per
    remote = torch.unique(cols[~mine])
    n_remote = int(remote.numel())
    new_cols = torch.where(mine, cols, torch.searchsorted(remote, cols) + per).to(torch.int32)
    compact = sparse.DeviceCSR(local.n_rows, per + n_remote, local.indptr, new_cols, local.values)
    del cols, mine, new_cols
    x_own = sparse.synthetic_dense(per, f, seed=7)
    x_kept = sparse.synthetic_dense(n_remote, f, seed=8)          # stands for the rows fetched once
    out = torch.empty((compact.n_rows, f), device=dev)
    # kept: two pack launches into one operand, one packed product
    packed = sparse.PackedRows.empty(per + n_remote, f, dev)

    def kept_form():
        sparse.dropout_pack(x_own, keep, 5, 1, out=packed.rows(0, per))
        sparse.dropout_pack(x_kept, keep, 5, 1, row_ids=remote, out=packed.rows(per, per + n_remote))
        sparse.spmm_packed(compact, packed, out=out)
    t_kept = timed(kept_form)
    t_kept_pack = timed(lambda: (sparse.dropout_pack(x_own, keep, 5, 1, out=packed.rows(0, per)),
                                 sparse.dropout_pack(x_kept, keep, 5, 1, row_ids=remote,
                                                     out=packed.rows(per, per + n_remote))))
    del packed
    torch.cuda.empty_cache()
    # sent: dense dropout of the own rows, pack what the others asked for (as many rows as this rank receives),
    # a copy for the links, the two-part dense product
    send_rows = torch.randint(0, per, (n_remote,), device=dev)
    send_buf = torch.empty((n_remote, f), device=dev)
    recv_buf = torch.empty((n_remote, f), device=dev)

    def sent_form():
        dropped = sparse.dropout(x_own, keep, 5, 1)
        sparse.gather_rows(dropped, send_rows, send_buf)
        recv_buf.copy_(send_buf)
        sparse.spmm(compact, recv_buf, out=out, first=dropped)
    t_sent = timed(sent_form)
    link_gb = n_remote * f * 4 / 1e9
    link_ms = link_gb / (LINK_GBPS * (world - 1)) * 1e3
    ideal = (t_pack1 + t_prod1) / world
    print('%d ranks: rank 0 forms %d rows (%d entries) from %d own + %d kept rows of X; kept form %.2f ms (packing %.2f); '
          'sent form %.2f ms + %.2f GB over %d links (%.2f ms at %.0f GB/s per link, not overlapped); one GPU / %d = %.2f ms'
          % (world, compact.n_rows, compact.nnz, per, n_remote, t_kept, t_kept_pack, t_sent, link_gb, world - 1, link_ms,
             LINK_GBPS, world, ideal), flush=True)
    results[str(world)] = {'rows': compact.n_rows, 'nnz': compact.nnz, 'own_rows': per, 'kept_rows': n_remote,
                           'kept_form_ms': t_kept, 'kept_form_pack_ms': t_kept_pack, 'sent_form_ms': t_sent,
                           'sent_form_link_gb': link_gb, 'sent_form_link_ms': link_ms, 'one_gpu_over_p_ms': ideal}
    del shard, local, compact, x_own, x_kept, out, send_rows, send_buf, recv_buf, remote
    torch.cuda.empty_cache()
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as fh:
        json.dump(results, fh, indent=1)
