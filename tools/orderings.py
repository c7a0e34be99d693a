#!/usr/bin/env python3
"""Where the traffic of A_hat . B goes: the same product on the R-MAT bench graph under three vertex
orders, with the hot-column cache hint, in column halves, and against dense operands that fit one cache
level (column indices folded modulo M).

    python tools/orderings.py --scale 24 --features 256 --order scrambled,natural,degree
    python tools/orderings.py --order degree --hot 0,4096,32768,262144
    python tools/orderings.py --order scrambled --fold 4096,32768,131072,1048576
    python tools/orderings.py --order scrambled --halves

Prints one JSON line per measurement.  Under rocprofv3 (--kernel-trace / --pmc) give ONE order and one
variant per run so that the launches of the row kernel belong to it.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse   # noqa: E402


def hat(indptr, indices):
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    return vals


def build(scale, order):
    """(DeviceCSR of A_hat, new_id or None): ``degree`` relabels the scrambled graph by descending degree."""
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True, scramble=(order != 'natural'))
    new_id = None
    if order.startswith('hubs'):
        # the K vertices of highest degree first (in degree order), every other vertex after them in its old order
        k = int(order[4:])
        by_degree, _ = sparse.degree_order(indptr)                     # position of every vertex in degree order
        is_hub = by_degree < k
        rest_rank = torch.cumsum((~is_hub).to(torch.int32), 0, dtype=torch.int32) - 1
        new_id = torch.where(is_hub, by_degree, rest_rank + k).to(torch.int32).contiguous()
        indptr, indices, _ = sparse.csr_relabel(n, n, indptr, indices, None, new_id, new_id)
        indices = indices.clone()
        del by_degree, is_hub, rest_rank
        torch.cuda.empty_cache()
    if order == 'degree':
        new_id, old_id = sparse.degree_order(indptr)
        indptr, indices, _ = sparse.csr_relabel(n, n, indptr, indices, None, new_id, new_id)
        indices = indices.clone()
        del old_id
        torch.cuda.empty_cache()
    vals = hat(indptr, indices)
    torch.cuda.empty_cache()
    return sparse.DeviceCSR(n, n, indptr, indices, vals), new_id


def time_product(a, b, c, reps, warmup=2, **kw):
    a.plan.set_timing(True)
    for _ in range(warmup):
        sparse.spmm(a, b, out=c, **kw)
    ms = []
    for _ in range(reps):
        sparse.spmm(a, b, out=c, **kw)
        ms.append(sum(a.plan.last_timing()))
    a.plan.set_timing(False)
    ms.sort()
    return ms[len(ms) // 2], ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=int, default=24)
    ap.add_argument('--features', type=int, default=256)
    ap.add_argument('--order', default='scrambled')
    ap.add_argument('--hot', default='0', help='comma list of hot-column counts (0 = hint off)')
    ap.add_argument('--fold', default='', help='comma list of M: column indices taken modulo M, B has M rows')
    ap.add_argument('--halves', action='store_true', help='also the product as two passes over 128 columns each')
    ap.add_argument('--lds-hubs', default='', help='comma list of hub-row counts staged in LDS (narrow widths)')
    ap.add_argument('--slabs', type=int, default=0,
                    help='also the product as passes over column tiles of this many columns (the narrow kernel), '
                         'without and with the --lds-hubs rows staged')
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    n, f = 1 << args.scale, args.features
    for order in args.order.split(','):
        a, _ = build(args.scale, order)
        b = sparse.synthetic_dense(n, f, seed=7)
        c = torch.empty_like(b)
        base = {'scale': args.scale, 'features': f, 'order': order, 'nnz': a.nnz,
                'long_rows': a.plan.n_long_rows}
        for hot in (int(h) for h in args.hot.split(',')):
            a.plan.set_hot_columns(hot)
            med, best = time_product(a, b, c, args.reps)
            print(json.dumps(dict(base, variant='full width', hot_cols=hot, ms_median=med, ms_min=best)), flush=True)
        a.plan.set_hot_columns(0)
        for hubs in (int(x) for x in args.lds_hubs.split(',') if x):
            a.plan.set_lds_hub_rows(hubs)
            med, best = time_product(a, b, c, args.reps)
            print(json.dumps(dict(base, variant='hub rows in LDS', lds_hub_rows=hubs, ms_median=med, ms_min=best)),
                  flush=True)
        a.plan.set_lds_hub_rows(0)
        if args.slabs:
            for hubs in [0] + [int(x) for x in args.lds_hubs.split(',') if x]:
                a.plan.set_lds_hub_rows(hubs)
                a.plan.set_timing(True)
                ms = []
                for _ in range(args.reps + 1):
                    t = 0.0
                    for c0 in range(0, f, args.slabs):
                        sparse.spmm(a, b[:, c0:c0 + args.slabs], out=c[:, c0:c0 + args.slabs])
                        t += sum(a.plan.last_timing())
                    ms.append(t)
                a.plan.set_timing(False)
                ms = sorted(ms[1:])
                print(json.dumps(dict(base, variant='column tiles of %d' % args.slabs, lds_hub_rows=hubs,
                                      ms_median=ms[len(ms) // 2], ms_min=ms[0])), flush=True)
            a.plan.set_lds_hub_rows(0)
        if args.halves:
            a.plan.set_timing(True)
            ms = []
            for _ in range(args.reps + 2):
                t = 0.0
                for c0 in range(0, f, f // 2):
                    sparse.spmm(a, b[:, c0:c0 + f // 2], out=c[:, c0:c0 + f // 2], wide_split=True)
                    t += sum(a.plan.last_timing())
                ms.append(t)
            a.plan.set_timing(False)
            ms = sorted(ms[2:])
            print(json.dumps(dict(base, variant='two column halves', ms_median=ms[len(ms) // 2], ms_min=ms[0])),
                  flush=True)
        for m in (int(x) for x in args.fold.split(',') if x):
            folded = sparse.DeviceCSR(n, m, a.indptr, (a.indices % m).to(torch.int32), a.values)
            bm = b[:m].contiguous()
            med, best = time_product(folded, bm, c, args.reps)
            print(json.dumps(dict(base, variant='columns folded modulo %d' % m, operand_mb=m * f * 4 / 1e6,
                                  ms_median=med, ms_min=best)), flush=True)
            del folded, bm
        del a, b, c
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
