"""The wide product with a packed operand beside the dense one, on the bench graph: A_hat (R-MAT scale 24) times
dropout(X) (half kept) and times a matrix with a quarter of its elements nonzero.

    python tools/packed_probe.py [--scale 24]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402


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
args = ap.parse_args()
n, f = 1 << args.scale, 256
indptr, indices = sparse.rmat_csr(args.scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
a = sparse.DeviceCSR(n, n, indptr, indices, vals)
x = sparse.synthetic_dense(n, f, seed=7)
c = torch.empty((n, f), device='cuda')
c2 = torch.empty((n, f), device='cuda')
print('dense product: %.2f ms' % timed(lambda: sparse.spmm(a, x, out=c)), flush=True)
for keep in (0.5, 0.25):
    dropped = sparse.dropout(x, keep, 5, 1)
    packed = sparse.PackedRows.empty(n, f, x.device)
    t_drop = timed(lambda: sparse.dropout(x, keep, 5, 1))
    t_pack = timed(lambda: sparse.dropout_pack(x, keep, 5, 1, out=packed))
    t_dense = timed(lambda: sparse.spmm(a, dropped, out=c))
    t_packed = timed(lambda: sparse.spmm_packed(a, packed, out=c2))
    sparse.spmm(a, dropped, out=c, wide_split=True)
    same = bool(torch.equal(c, c2))
    print('keep %.2f: dropout pass %.2f ms, dropout into packed rows %.2f ms; product dense %.2f ms, packed %.2f ms; '
          'equal %s' % (keep, t_drop, t_pack, t_dense, t_packed, same), flush=True)
    del dropped, packed
    torch.cuda.empty_cache()
