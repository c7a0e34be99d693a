"""A few launches of the product A_hat . B on the R-MAT bench graph at a given width, for rocprofv3
(kernel trace and counter passes):  python tools/spmm_probe.py --scale 24 --features 16"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=24)
ap.add_argument('--features', type=int, default=16)
ap.add_argument('--reps', type=int, default=5)
args = ap.parse_args()
n = 1 << args.scale
indptr, indices = sparse.rmat_csr(args.scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
a = sparse.DeviceCSR(n, n, indptr, indices, vals)
b = sparse.synthetic_dense(n, args.features, seed=7)
c = torch.empty_like(b)
for _ in range(args.reps):
    sparse.spmm(a, b, out=c)
torch.cuda.synchronize()
print('done', a.nnz)
