#!/usr/bin/env python3
"""Five launches of gcn_gemm_f32 on 2^22 x 512 x 512 (or m k n from the command line) for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402

m, k, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1 << 22, 512, 512)
a = sparse.synthetic_dense(m, k, seed=3)
b = sparse.synthetic_dense(k, n, seed=4)
c = torch.empty((m, n), dtype=torch.float32, device='cuda')
for _ in range(5):
    sparse.gemm(a, b, out=c)
torch.cuda.synchronize()
