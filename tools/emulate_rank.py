"""One rank's share of the row-sharded product, measured on ONE GPU through the code the ranks run.

Rank 0's shard of the scale-24 R-MAT graph for 2, 4 and 8 ranks goes through ``dist.halo_product`` with a
plan whose exchange is a device copy of the bytes the rank would receive (``HaloPlan`` without a
communicator): packing on the plan's side stream, the copy standing in for the link transfer (it loads
HBM as sending plus receiving does, it does not take the time a link takes), the products of the column
ranges on the caller's stream, reading the rank's own rows of B in place and the received rows from the
exchange buffer.  Measured per candidate split: the product of every range alone, the whole pipeline, the
packing and the copy of every range.  Modelled: the timeline of a step in which range i crosses the links
after range i - 1 at the nominal 153 GB/s per link (P - 1 links in use) and its product starts once it
has arrived and the product before it has finished -- the products slowed down by the factor the measured
pipeline shows them to lose beside the exchange traffic -- and the strong-scaling efficiency
E(P) = T_1 / (P T_P) against the one-GPU product measured in the same run.

    python tools/emulate_rank.py [--scale 24] [--features 256] [--out gpurun_out/r3/rank_emulation.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse, dist as gdist   # noqa: E402

LINK_GBPS = 153.0
LINK_RATES = (153.0, 100.0, 64.0, 48.0)      # GB/s per link and direction the timeline is also evaluated at


def timed(fn, reps=8):
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
    return float(np.median(ms))


def stand_in_plan(shard, per):
    """A HaloPlan of one process standing for one rank of ``world``: the compact matrix (own rows first, then
    the rows of other blocks the shard references) and the row counts of the real plan; the rows it packs are
    random rows of its own block, as many as it receives, and the "exchange" copies them into the receive
    buffer."""
    local = shard.local
    r0 = shard.row_begin
    cols = local.indices.to(torch.int64)
    mine = (cols >= r0) & (cols < r0 + per)
    remote = torch.unique(cols[~mine])
    new_cols = torch.where(mine, cols - r0, torch.searchsorted(remote, cols) + per).to(torch.int32)
    plan = gdist.HaloPlan.__new__(gdist.HaloPlan)
    plan.shard, plan.group = shard, None
    plan.n_first, plan.n_remote = per, int(remote.numel())
    plan.compact = sparse.DeviceCSR(local.n_rows, per + plan.n_remote, local.indptr, new_cols, local.values)
    plan._identity, plan._collective, plan._buffers, plan._side = False, False, {}, None
    plan.own_first, plan.own, plan.remote = False, None, None
    plan.send_rows = torch.randint(0, per, (plan.n_remote,), device=local.device)
    referenced_own = int(torch.unique(cols[mine]).numel())
    return plan, referenced_own


def own_and_remote(plan):
    """The compact matrix cut by column: the entries that read the rank's own rows, and the entries that read
    received rows (columns renumbered from 0), entry order inside a row unchanged."""
    a = plan.compact
    own = a.indices < plan.n_first
    rows = torch.repeat_interleave(torch.arange(a.n_rows, device=a.device), a.indptr[1:] - a.indptr[:-1])

    def part(keep, shift, n_cols):
        counts = torch.bincount(rows[keep], minlength=a.n_rows)
        indptr = torch.zeros(a.n_rows + 1, dtype=torch.int64, device=a.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        return sparse.DeviceCSR(a.n_rows, n_cols, indptr, (a.indices[keep] - shift).contiguous(), a.values[keep].contiguous())

    return part(own, 0, plan.n_first), part(~own, plan.n_first, plan.n_remote)


def own_first(plan, a_own, a_remote, b_local, out, ranges):
    """The product with the part that needs no exchange first: every range's exchange is issued, the own-rows
    product of the full width runs while they are on the links, then each range's received rows are multiplied
    and added to what is there."""
    plan.own_first, plan.own, plan.remote = True, a_own, a_remote
    try:
        return gdist.halo_product(plan, b_local, out, False, ranges)
    finally:
        plan.own_first = False


def own_first_report(world, plan, b_local, out, halo, t1, links, recv_rows, f):
    a_own, a_remote = own_and_remote(plan)
    t_own = timed(lambda: sparse.spmm(a_own, b_local, out=out))
    print('world %d own-first: own part %d entries %.2f ms, remote part %d entries' % (world, a_own.nnz, t_own, a_remote.nnz),
          flush=True)
    rows = []
    for shares in gdist.HaloSpmm.CANDIDATES:
        ranges = gdist.column_shares(f, shares)
        widths = [hi - lo for lo, hi in ranges]
        t_each = [timed(lambda lo=lo, hi=hi: sparse.spmm_add(a_remote, halo[:, lo:hi], out[:, lo:hi], out=out[:, lo:hi],
                                                             wide_split=f >= 128)) for lo, hi in ranges]
        t_pipe = timed(lambda: own_first(plan, a_own, a_remote, b_local, out, ranges))
        bufs = [plan._buffers_for(w, i, b_local.device) for i, w in enumerate(widths)]
        t_pack0 = timed(lambda: sparse.gather_rows(b_local[:, ranges[0][0]:ranges[0][1]], plan.send_rows, bufs[0][0]))
        link = [recv_rows * w * 4 / (links * LINK_GBPS * 1e9) * 1e3 for w in widths]
        slowdown = max(1.0, t_pipe / (t_own + sum(t_each)))
        sent, done, arrivals = t_pack0, t_own * slowdown, []
        for i in range(len(ranges)):
            sent += link[i]
            arrivals.append(sent)
            done = max(done, sent) + t_each[i] * slowdown
        at_rate = {}
        for rate in LINK_RATES:
            sent_r, done_r = t_pack0, t_own * slowdown
            for i in range(len(ranges)):
                sent_r += link[i] * LINK_GBPS / rate
                done_r = max(done_r, sent_r) + t_each[i] * slowdown
            at_rate[str(rate)] = t1 / (world * done_r)
        rows.append({'widths': widths, 'own_ms': t_own, 'remote_ms': t_each, 'pipeline_ms': t_pipe, 'slowdown': slowdown,
                     'arrival_ms': arrivals, 'modelled_step_ms': done, 'modelled_efficiency': t1 / (world * done),
                     'efficiency_at_link_gbps': at_rate})
        print('world %d own-first widths %-28s own %.2f + remote %s = %.2f  pipeline %.2f  slowdown %.3f  arrivals %s -> T_P %.2f ms, E = %.3f'
              % (world, widths, t_own, ['%.2f' % t for t in t_each], t_own + sum(t_each), t_pipe, slowdown,
                 ['%.2f' % t for t in arrivals], done, t1 / (world * done)), flush=True)
        plan.drop_buffers()
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=int, default=24)
    ap.add_argument('--features', type=int, default=256)
    ap.add_argument('--out', default=None)
    ap.add_argument('--own-first', action='store_true', help='only the form that multiplies the own rows first')
    args = ap.parse_args()
    scale, f = args.scale, args.features
    n = 1 << scale
    indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
    d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
    vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
    whole = sparse.DeviceCSR(n, n, indptr, indices, vals)
    b = sparse.synthetic_dense(n, f, seed=7)
    c = torch.empty((n, f), device='cuda')
    t1 = timed(lambda: sparse.spmm(whole, b, out=c))
    del c, whole
    torch.cuda.empty_cache()
    print('one GPU, whole product: %.2f ms' % t1, flush=True)
    report = {'scale': scale, 'features': f, 'one_gpu_ms': t1, 'link_gbps': LINK_GBPS, 'worlds': []}
    for world in (2, 4, 8):
        per = n // world
        shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, world)
        plan, own = stand_in_plan(shard, per)
        b_local = b[:per]
        out = torch.empty((per, f), device='cuda')
        recv_rows = plan.n_remote                          # rows that come over the links
        links = world - 1
        halo = torch.empty((plan.n_remote, f), device='cuda').copy_(b[per:per + plan.n_remote])
        t_full = timed(lambda: sparse.spmm(plan.compact, halo, out=out, first=b_local))
        entry = {'world': world, 'rows_per_rank': per, 'referenced_rows': own + plan.n_remote,
                 'referenced_share_of_n': (own + plan.n_remote) / n, 'rows_over_links': recv_rows,
                 'received_gb_per_product': recv_rows * f * 4 / 1e9, 'local_product_full_width_ms': t_full,
                 'link_total_ms': recv_rows * f * 4 / (links * LINK_GBPS * 1e9) * 1e3, 'splits': []}
        if args.own_first:
            entry['own_first'] = own_first_report(world, plan, b_local, out, halo, t1, links, recv_rows, f)
        for shares in (() if args.own_first else gdist.HaloSpmm.CANDIDATES):
            ranges = gdist.column_shares(f, shares)
            widths = [hi - lo for lo, hi in ranges]

            def product_of(i):
                lo, hi = ranges[i]
                return lambda: sparse.spmm(plan.compact, halo[:, lo:hi], out=out[:, lo:hi], wide_split=f >= 128,
                                           first=b_local[:, lo:hi])

            def pipeline():
                if len(ranges) > 1:
                    gdist.halo_product(plan, b_local, out, False, ranges)
                else:                              # one range: packing, the whole exchange, then the product
                    plan.multiply(b_local, plan.exchange(b_local), out=out)

            t_each = [timed(product_of(i)) for i in range(len(ranges))]
            t_pipe = timed(pipeline)
            bufs = [plan._buffers_for(w, i, b_local.device) for i, w in enumerate(widths)]
            t_pack = [timed(lambda i=i: sparse.gather_rows(b_local[:, ranges[i][0]:ranges[i][1]], plan.send_rows, bufs[i][0]))
                      for i in range(len(ranges))]
            t_copy = [timed(lambda i=i: bufs[i][1].copy_(bufs[i][0])) for i in range(len(ranges))]
            link = [recv_rows * w * 4 / (links * LINK_GBPS * 1e9) * 1e3 for w in widths]
            # How much the products slow down while the exchange traffic shares the memory system with them:
            # the measured pipeline against its own critical path (first pack and copy, then the products).
            slowdown = max(1.0, (t_pipe - t_pack[0] - t_copy[0]) / sum(t_each))
            # Timeline of a step on P GPUs: range i is packed (packing of range i + 1 overlaps the transfer of
            # range i), crosses the links one range after the other, and its product starts when it has arrived
            # and the product before it has finished.
            sent, done, arrivals = t_pack[0], 0.0, []
            for i in range(len(ranges)):
                sent += link[i]
                arrivals.append(sent)
                done = max(done, sent) + t_each[i] * slowdown
            modelled = done
            # the same timeline at other link rates (what a collective achieves per link is below the nominal rate)
            at_rate = {}
            for rate in LINK_RATES:
                sent_r, done_r = t_pack[0], 0.0
                for i in range(len(ranges)):
                    sent_r += link[i] * LINK_GBPS / rate
                    done_r = max(done_r, sent_r) + t_each[i] * slowdown
                at_rate[str(rate)] = t1 / (world * done_r)
            entry['splits'].append({'widths': widths, 'product_ms': t_each, 'pipeline_ms': t_pipe, 'pack_ms': t_pack,
                                    'copy_ms': t_copy, 'link_ms': link, 'arrival_ms': arrivals,
                                    'slowdown_beside_exchange': slowdown, 'modelled_step_ms': modelled,
                                    'modelled_efficiency': t1 / (world * modelled), 'efficiency_at_link_gbps': at_rate})
            print('world %d widths %-20s products %s = %.2f  pipeline %.2f  slowdown %.3f  arrivals %s -> T_P %.2f ms, E = %.3f'
                  % (world, widths, ['%.2f' % t for t in t_each], sum(t_each), t_pipe, slowdown,
                     ['%.2f' % t for t in arrivals], modelled, t1 / (world * modelled)), flush=True)
            plan.drop_buffers()
            torch.cuda.empty_cache()
        if args.own_first:
            best = min(entry['own_first'], key=lambda s_: s_['modelled_step_ms'])
            print('world %d own-first best %s: E = %.3f' % (world, best['widths'], best['modelled_efficiency']), flush=True)
            for rate in LINK_RATES:
                top = max(entry['own_first'], key=lambda s_: s_['efficiency_at_link_gbps'][str(rate)])
                print('world %d own-first at %.0f GB/s per link: best %s E = %.3f'
                      % (world, rate, top['widths'], top['efficiency_at_link_gbps'][str(rate)]), flush=True)
            report['worlds'].append(entry)
            del shard, plan, halo, out, b_local
            torch.cuda.empty_cache()
            continue
        best = min(entry['splits'], key=lambda s_: s_['modelled_step_ms'])
        entry['best'] = best
        for rate in LINK_RATES:
            top = max(entry['splits'], key=lambda s_: s_['efficiency_at_link_gbps'][str(rate)])
            print('world %d in order at %.0f GB/s per link: best %s E = %.3f'
                  % (world, rate, top['widths'], top['efficiency_at_link_gbps'][str(rate)]), flush=True)
        report['worlds'].append(entry)
        print('world %d: references %.1f%% of N, full-width local product %.2f ms, links busy %.2f ms per product; '
              'best split %s: E(%d) = %.3f' % (world, 100.0 * (own + plan.n_remote) / n, t_full, entry['link_total_ms'],
                                               best['widths'], world, best['modelled_efficiency']), flush=True)
        del shard, plan, halo, out, b_local
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as fh:
            json.dump(report, fh, indent=1)


if __name__ == '__main__':
    main()
