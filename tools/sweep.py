"""Measurement sweeps on one MI355X, written as a markdown report (committed under profiles/).

  python tools/sweep.py --out gpurun_out/sweeps.md [--quick]

Sections: feature-width sweep of the product on R-MAT scale 22 (F up to 512) and scale 24
(F up to 256); A_hat^k X on the PubMed graph, launched step by step and as one HIP graph;
the dense transform (gcn_gemm_f32) on the shapes of the synthetic configurations.  Timing is
HIP events around >= 20 repetitions after 3 warm-ups, median and minimum, buffers allocated once.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gcn_amd import smoothing, sparse          # noqa: E402

HBM_PEAK = 8.0e12
MFMA_F32_PEAK = 157.3e12


def timed(fn, reps=20, warmup=3):
    for _ in range(warmup):
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


def rmat_support(scale):
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    n = 1 << scale
    return sparse.DeviceCSR(n, n, indptr, indices, vals)


def width_sweep(out, scale, widths, reps):
    a = rmat_support(scale)
    n, nnz = a.n_rows, a.nnz
    out.append('\n## Product `A_hat . B`, R-MAT scale %d (N = %d, nnz = %d)\n' % (scale, n, nnz))
    out.append('| F | median ms | min ms | bytes_min GB/s | % of 8 TB/s | gathered GB/s | GFLOP/s |')
    out.append('|---|---|---|---|---|---|---|')
    for f in widths:
        b = sparse.synthetic_dense(n, f, seed=7)
        c = torch.empty_like(b)
        med, mn = timed(lambda: sparse.spmm(a, b, out=c), reps=reps)
        bmin = 8 * nnz + 8 * (n + 1) + 8 * n * f
        bgat = 8 * nnz + 8 * (n + 1) + 4 * f * nnz + 4 * n * f
        out.append('| %d | %.2f | %.2f | %.0f | %.1f | %.0f | %.0f |' % (
            f, med, mn, bmin / med / 1e6, 100 * bmin / (med * 1e-3) / HBM_PEAK, bgat / med / 1e6,
            2.0 * nnz * f / med / 1e6))
        del b, c
    del a
    torch.cuda.empty_cache()


def pubmed_powers(out, reps):
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'pubmed_graph.npz'))
    n = int(fx['n'])
    a = sparse.DeviceCSR(n, n, torch.as_tensor(fx['ai_indptr'].astype(np.int64)).cuda(),
                         torch.as_tensor(fx['ai_indices'].astype(np.int32)).cuda(),
                         torch.as_tensor(fx['hat_values'].astype(np.float32)).cuda())
    f = 500
    x = sparse.synthetic_dense(n, f, seed=7)
    bmin = 8 * a.nnz + 8 * (n + 1) + 8 * n * f
    out.append('\n## `A_hat^k X` on the PubMed graph (N = %d, nnz = %d, X %d x %d; %.1f MB per step, cache-resident)\n'
               % (n, a.nnz, n, f, bmin / 1e6))
    out.append('| k | step-by-step ms | per step us | one HIP graph ms | per step us | per-step GB/s (graph) |')
    out.append('|---|---|---|---|---|---|')
    for k in (1, 2, 4, 8, 16, 32, 64):
        eager, _ = timed(lambda: smoothing.propagate(a, x, k), reps=reps)
        g = smoothing.GraphedPropagation(a, f, k)
        g.x.copy_(x)
        graphed, _ = timed(lambda: g.graph.replay(), reps=reps)
        out.append('| %d | %.3f | %.1f | %.3f | %.1f | %.0f |' % (
            k, eager, 1e3 * eager / k, graphed, 1e3 * graphed / k, bmin / (graphed / k) / 1e6))
        del g
    # the same steps with a 512 MB memset between repetitions: operands come from HBM, not the cache
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    c = torch.empty_like(x)

    def cold():
        flush.zero_()
        sparse.spmm(a, x, out=c)
    both, _ = timed(cold, reps=reps)
    only_flush, _ = timed(lambda: flush.zero_(), reps=reps)
    out.append('\nCold single step (512 MB memset before every repetition): %.1f us (memset alone %.1f us).'
               % (1e3 * (both - only_flush), 1e3 * only_flush))


def gemm_shapes(out, reps, quick):
    out.append('\n## Dense transform `gcn_gemm_f32` (f32 MFMA, peak 157.3 TFLOP/s)\n')
    out.append('| M | K | N | B form | median ms | TFLOP/s | % of peak | GB/s (A+B+C once) | torch.matmul ms |')
    out.append('|---|---|---|---|---|---|---|---|---|')
    shapes = [(1 << 22, 512, 512, False), (1 << 22, 512, 512, True), (8192, 8192, 8192, False),
              (1 << 24, 256, 16, False), (1 << 24, 16, 8, False), (1 << 22, 512, 16, False),
              (1 << 20, 256, 256, False)]
    if quick:
        shapes = shapes[-2:]
    for m, k, n, trans_b in shapes:
        a = sparse.synthetic_dense(m, k, seed=3)
        b = sparse.synthetic_dense(n, k, seed=4) if trans_b else sparse.synthetic_dense(k, n, seed=4)
        c = torch.empty((m, n), dtype=torch.float32, device='cuda')
        med, _ = timed(lambda: sparse.gemm(a, b, trans_b=trans_b, out=c), reps=reps)
        bt = b.t() if trans_b else b
        ref, _ = timed(lambda: torch.matmul(a, bt, out=c), reps=reps)
        flop = 2.0 * m * n * k
        byt = 4.0 * (m * k + k * n + m * n)
        out.append('| %d | %d | %d | %s | %.3f | %.1f | %.1f | %.0f | %.3f |' % (
            m, k, n, 'N x K (transposed)' if trans_b else 'K x N', med, flop / med / 1e9,
            100 * flop / (med * 1e-3) / MFMA_F32_PEAK, byt / med / 1e6, ref))
        del a, b, c
        torch.cuda.empty_cache()


def weight_gradient_shapes(out, reps, quick):
    out.append('\n## Weight gradient `gcn_gemm_tn_f32` (C[k x n] = A^T . G, sum over m nodes; HBM-bound: A and G once)\n')
    out.append('| m | k | n | median ms | GB/s (A+G once) | % of 8 TB/s | TFLOP/s | torch.matmul(A.t(), G) ms |')
    out.append('|---|---|---|---|---|---|---|---|')
    shapes = [(1 << 24, 256, 256), (1 << 24, 256, 16), (1 << 24, 16, 8), (1 << 22, 512, 512), (1 << 22, 512, 16)]
    if quick:
        shapes = [(1 << 18, 256, 16)]
    for m, k, n in shapes:
        a = sparse.synthetic_dense(m, k, seed=3)
        g = sparse.synthetic_dense(m, n, seed=4)
        c = torch.empty((k, n), dtype=torch.float32, device='cuda')
        med, _ = timed(lambda: sparse.gemm_tn(a, g, out=c), reps=reps)
        ref, _ = timed(lambda: torch.matmul(a.t(), g, out=c), reps=reps)
        byt = 4.0 * m * (k + n)
        out.append('| %d | %d | %d | %.3f | %.0f | %.1f | %.1f | %.3f |' % (
            m, k, n, med, byt / med / 1e6, 100 * byt / (med * 1e-3) / HBM_PEAK, 2.0 * m * n * k / med / 1e9, ref))
        del a, g, c
        torch.cuda.empty_cache()


def sparse_feature_layer(out, reps, quick):
    out.append('\n## First layer on sparse features: `sparse_dropout(X) . W0` fused (bag-of-words X, 50 entries per row, 4096 columns; W0 4096 x F)\n')
    out.append('| rows | nnz(X) | F | dropout | median ms | CSR bytes GB/s (8 nnz + 8 N + 4 N F) | gathered GB/s |')
    out.append('|---|---|---|---|---|---|---|')
    n = 1 << (16 if quick else 22)
    per_row, f_in = 50, 4096
    gen = torch.Generator(device='cuda').manual_seed(3)
    indices = torch.randint(0, f_in, (n, per_row), generator=gen, device='cuda', dtype=torch.int32).sort(dim=1).values.reshape(-1)
    indptr = torch.arange(n + 1, device='cuda', dtype=torch.int64) * per_row
    vals = torch.rand(n * per_row, generator=gen, device='cuda')
    x = sparse.DeviceCSR(n, f_in, indptr, indices, vals)
    for f in (16, 64):
        w = sparse.synthetic_dense(f_in, f, seed=5)
        c = torch.empty((n, f), dtype=torch.float32, device='cuda')
        for keep in (1.0, 0.5):
            if keep < 1.0:
                med, _ = timed(lambda: sparse.spmm_dropout(x, w, keep, 7, 3, out=c), reps=reps)
            else:
                med, _ = timed(lambda: sparse.spmm(x, w, out=c), reps=reps)
            byt = 8.0 * x.nnz + 8.0 * n + 4.0 * n * f
            gat = 8.0 * x.nnz + 8.0 * n + 4.0 * f * x.nnz + 4.0 * n * f
            out.append('| %d | %d | %d | %s | %.3f | %.0f | %.0f |' % (n, x.nnz, f, 'keep 0.5' if keep < 1 else 'off', med,
                                                                     byt / med / 1e6, gat / med / 1e6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/sweeps.md')
    ap.add_argument('--quick', action='store_true', help='small scales only (smoke run)')
    ap.add_argument('--sections', default='widths,powers,gemm,wgrad,features')
    args = ap.parse_args()
    reps = 5 if args.quick else 20
    out = ['# Sweeps on one MI355X (%s)' % time.strftime('%Y-%m-%d'), '',
           'HIP-event timing, %d repetitions after 3 warm-ups, inputs resident in HBM.' % reps]
    sections = args.sections.split(',')
    if 'widths' in sections:
        if args.quick:
            width_sweep(out, 16, [16, 256], reps)
        else:
            width_sweep(out, 22, [8, 16, 32, 64, 128, 256, 512], reps)
            width_sweep(out, 24, [16, 32, 64, 128, 256], reps)
        print('widths done', flush=True)
    if 'powers' in sections:
        pubmed_powers(out, reps)
        print('powers done', flush=True)
    if 'gemm' in sections:
        gemm_shapes(out, reps, args.quick)
        print('gemm done', flush=True)
    if 'features' in sections:
        sparse_feature_layer(out, reps, args.quick)
        print('features done', flush=True)
    if 'wgrad' in sections:
        weight_gradient_shapes(out, reps, args.quick)
        print('wgrad done', flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'w') as fh:
        fh.write('\n'.join(out) + '\n')
    print('\n'.join(out))


if __name__ == '__main__':
    main()
