"""gcn_gemm_f32 on shapes with many column tiles, with the block order of the whole-panel kernel chosen by
GCN_GEMM_RASTER (0: the column tiles of one panel after the other; 1: groups of 8 panels per column tile).

    GCN_GEMM_RASTER=0 python tools/gemm_raster.py ; GCN_GEMM_RASTER=1 python tools/gemm_raster.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402


def timed(fn, reps=11, warmup=3):
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
    return ms[len(ms) // 2], ms[0]


for m, k, n in ((8192, 8192, 8192), (16384, 4096, 4096), (1 << 20, 512, 2048), (1 << 22, 512, 512), (1 << 24, 256, 256)):
    a = sparse.synthetic_dense(m, k, seed=3)
    b = sparse.synthetic_dense(k, n, seed=4)
    c = torch.empty((m, n), dtype=torch.float32, device='cuda')
    med, low = timed(lambda: sparse.gemm(a, b, out=c))
    ref = torch.matmul(a, b)
    err = float((c - ref).abs().max() / ref.abs().max())
    lib, lib_low = timed(lambda: torch.matmul(a, b, out=c))
    flop = 2.0 * m * n * k
    print('raster=%s  %d x %d x %d: %.3f ms (min %.3f) = %.1f TF/s (best %.1f)   torch.matmul %.1f TF/s   max rel diff %.1e' % (
        os.environ.get('GCN_GEMM_RASTER', 'default'), m, k, n, med, low, flop / med / 1e9, flop / low / 1e9,
        flop / lib / 1e9, err), flush=True)
    del a, b, c, ref
    torch.cuda.empty_cache()
