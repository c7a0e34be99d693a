"""A few launches of gcn_gemm_f32 on the dense-transform shape of the synthetic configuration
(2^22 x 512 . 512 x 512), for rocprofv3 (kernel trace and counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402

m, k, n = 1 << 22, 512, 512
a = sparse.synthetic_dense(m, k, seed=3)
b = sparse.synthetic_dense(k, n, seed=4)
c = torch.empty((m, n), dtype=torch.float32, device='cuda')
for _ in range(5):
    sparse.gemm(a, b, out=c)
torch.cuda.synchronize()
print('done')
