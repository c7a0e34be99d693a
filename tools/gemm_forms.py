"""Times of the dense products of a training step on the bench graph's shapes, each form beside the plain
product: the store with dropout, the left operand with dropout from mask bits, the weight gradients, the bit
generation and the stand-alone dropout pass they replace.  One JSON line per measurement.

    python tools/gemm_forms.py [--rows 16777216] [--k 256] [--n 256]
"""
import argparse
import json
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
    return float(np.median(ms)), float(np.min(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1 << 24)
    ap.add_argument('--k', type=int, default=256)
    ap.add_argument('--n', type=int, default=256)
    args = ap.parse_args()
    m, k, n = args.rows, args.k, args.n
    a = sparse.synthetic_dense(m, k, seed=1)
    b = sparse.synthetic_dense(k, n, seed=2) * 0.1
    g = sparse.synthetic_dense(m, n, seed=3)
    c = torch.empty((m, n), device='cuda')
    dw = torch.empty((k, n), device='cuda')
    bits = sparse.dropout_bits(m, k, 0.5, 11, 3)
    flop = 2.0 * m * n * k
    drop = sparse.OutDropout(0.5, 11, 4)
    forms = [
        ('gemm', lambda: sparse.gemm(a, b, out=c), flop),
        ('gemm relu', lambda: sparse.gemm(a, b, relu=True, out=c), flop),
        ('gemm relu + dropout in the store', lambda: sparse.gemm(a, b, relu=True, out=c, out_dropout=drop), flop),
        ('gemm, left operand dropped from mask bits', lambda: sparse.gemm_masked(a, bits, 0.5, b, out=c), flop),
        ('dropout bits', lambda: sparse.dropout_bits(m, k, 0.5, 11, 3), None),
        ('stand-alone dropout pass', lambda: sparse.dropout(a, 0.5, 11, 3), None),
        ('weight gradient', lambda: sparse.gemm_tn(a, g, out=dw), flop),
        ('weight gradient, left operand dropped from mask bits', lambda: sparse.gemm_tn_masked(a, bits, 0.5, g, out=dw), flop),
        ('torch.matmul', lambda: torch.matmul(a, b, out=c), flop),
    ]
    for name, fn, work in forms:
        med, low = timed(fn)
        rec = {'form': name, 'm': m, 'n': n, 'k': k, 'ms_median': med, 'ms_min': low}
        if work:
            rec['tflops'] = work / (med * 1e-3) / 1e12
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
