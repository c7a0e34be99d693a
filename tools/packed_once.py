"""A few launches of the packed-operand product (and of the dense one) on the bench graph, for rocprofv3:
python tools/packed_once.py [--keep 0.5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=24)
ap.add_argument('--keep', type=float, default=0.5)
args = ap.parse_args()
n, f = 1 << args.scale, 256
indptr, indices = sparse.rmat_csr(args.scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
a = sparse.DeviceCSR(n, n, indptr, indices, vals)
x = sparse.synthetic_dense(n, f, seed=7)
packed = sparse.dropout_pack(x, args.keep, 5, 1)
c = torch.empty((n, f), device='cuda')
for _ in range(4):
    sparse.spmm_packed(a, packed, out=c)
torch.cuda.synchronize()
