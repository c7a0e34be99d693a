"""The HBM-bound dense products of a GCN step with a 16-class output on 16.7 M nodes: hidden -> classes
(16.7M x 256 . 256 x 16), its weight gradient (256 x 16 from 16.7M rows) and the gated gradient with respect
to the hidden activations (16.7M x 16 . 16 x 256, gate read and output written).  Bytes moved / time beside each.

    python tools/narrow_gemm.py [--lib path/to/libgcn_hip.so]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import _lib, sparse   # noqa: E402

if '--lib' in sys.argv:                      # an experimental build of the library instead of the in-tree one
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])


def timed(fn, reps=9, warmup=2):
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


m, hidden, classes = 1 << 24, 256, 16
h = sparse.synthetic_dense(m, hidden, seed=3)
h = torch.where(h > 0, h, torch.zeros_like(h))           # a gate: positive where the unit was active and kept
w = sparse.synthetic_dense(hidden, classes, seed=4)
dp = sparse.synthetic_dense(m, classes, seed=5)
p = torch.empty((m, classes), device='cuda')
dh = torch.empty((m, hidden), device='cuda')
tag = os.path.basename(_lib.LIB_PATH)
t = timed(lambda: sparse.gemm(h, w, out=p))
print('%s  H.W1   (2^24 x 256 . 256 x 16): %.3f ms = %.2f TB/s' % (tag, t, (m * (hidden + classes) * 4) / t / 1e9), flush=True)
ref = (h[:4096].double() @ w.double())
if '--no-check' not in sys.argv:
    assert float((p[:4096].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
t = timed(lambda: sparse.gemm_tn(h, dp))
print('%s  H^T.dP (256 x 16 from 2^24 rows): %.3f ms = %.2f TB/s' % (tag, t, (m * (hidden + classes) * 4) / t / 1e9), flush=True)
t = timed(lambda: sparse.gemm_gated(dp, w, h, 0.5, trans_b=True, out=dh))
print('%s  (dP.W1^T) gated (2^24 x 16 . 16 x 256): %.3f ms = %.2f TB/s' % (tag, t, (m * (2 * hidden + classes) * 4) / t / 1e9), flush=True)
