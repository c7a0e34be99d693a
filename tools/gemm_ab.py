"""Time gcn_gemm_f32 on the dense-transform shapes (median of reps, HIP events), beside torch.matmul."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import _lib, sparse   # noqa: E402

if '--lib' in sys.argv:                      # an experimental build of the library instead of the in-tree one
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])


def timed(fn, reps=7, warmup=2):
    for _ in range(warmup):
        fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]


for m, k, n in ((1 << 22, 512, 512), (1 << 24, 256, 256), (8192, 8192, 8192)):
    a = sparse.synthetic_dense(m, k, seed=3)
    b = sparse.synthetic_dense(k, n, seed=4)
    c = torch.empty((m, n), dtype=torch.float32, device='cuda')
    ours = timed(lambda: sparse.gemm(a, b, out=c))
    lib = timed(lambda: torch.matmul(a, b, out=c)) if '--no-lib' not in sys.argv else float('nan')
    flop = 2.0 * m * n * k
    print('%d x %d x %d: ours %.3f ms = %.1f TF/s   torch.matmul %.3f ms = %.1f TF/s' % (
        m, k, n, ours, flop / ours / 1e9, lib, flop / lib / 1e9), flush=True)
    del a, b, c
    torch.cuda.empty_cache()

# the weight gradient: C[k x n] = A^T . G, reduction over the m rows
for m, k, n in ((1 << 24, 256, 256), (1 << 22, 512, 512), (1 << 21, 256, 256)):
    a = sparse.synthetic_dense(m, k, seed=3)
    g = sparse.synthetic_dense(m, n, seed=5)
    ours = timed(lambda: sparse.gemm_tn(a, g))
    lib = timed(lambda: torch.matmul(a.t(), g)) if '--no-lib' not in sys.argv else float('nan')
    flop = 2.0 * m * n * k
    print('A^T.G %d rows, %d x %d: ours %.3f ms = %.1f TF/s   torch.matmul %.3f ms = %.1f TF/s' % (
        m, k, n, ours, flop / ours / 1e9, lib, flop / lib / 1e9), flush=True)
    del a, g
    torch.cuda.empty_cache()
