"""The three pack kernels on 2^24 rows of 256 columns: dropout into packed rows, the nonzeros of a matrix, and the
gated gradient of a hidden layer written packed from its small product.

    python tools/pack_probe.py
"""
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


n, f, k = 1 << 24, 256, 16
x = sparse.synthetic_dense(n, f, seed=7)
packed = sparse.PackedRows.empty(n, f, x.device)
for keep in (0.5, 0.25):
    print('dropout(x, %.2f) into packed rows: %.2f ms' % (keep, timed(lambda: sparse.dropout_pack(x, keep, 5, 1, out=packed))),
          flush=True)
gate = sparse.dropout(torch.clamp(x, min=0), 0.5, 5, 2)          # a quarter positive, the rest zero
print('nonzeros of a quarter-dense matrix into packed rows: %.2f ms' % timed(lambda: sparse.pack_rows(gate, out=packed)),
      flush=True)
dp = sparse.synthetic_dense(n, k, seed=8)
w = sparse.synthetic_dense(f, k, seed=9)
print('gated product (k = %d) into packed rows: %.2f ms' % (k, timed(lambda: sparse.gemm_gated_pack(dp, w, gate, 0.5, out=packed))),
      flush=True)
# the same with seven rows of the small operand in eight all zero (the gradient of a hidden layer is non-zero only
# at the neighbours of labelled nodes) and the live rows listed: the other rows are emptied without being read
live = torch.arange(0, n, 8, device=dp.device)
dp_sparse = torch.zeros_like(dp)
dp_sparse[live] = dp[live]
print('gated product, 1 row in 8 listed as live: %.2f ms'
      % timed(lambda: sparse.gemm_gated_pack(dp_sparse, w, gate, 0.5, out=packed, rows=live)), flush=True)
