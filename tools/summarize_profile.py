#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into one small text file.

usage: summarize_profile.py <dir with trace/ pmc_*/ sub-directories> <out.md>
Only kernels of this library (namespace gcn::) are listed by full name; the rest are
abbreviated.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; the corrected
HBM byte count follows the MI355X rule: a wide coalesced read stream is counted at half
its bytes, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.strip('"')
    if 'gcn::' in name:
        i = name.index('gcn::')
        j = name.find('(', i)
        return name[i:j if j > 0 else None]
    if 'rocprim' in name:
        for key in ('radix_sort_onesweep_iteration', 'radix_sort_onesweep_global_offsets', 'scan_impl',
                    'init_lookback_scan_state'):
            if key in name:
                return 'rocprim::' + key
        return 'rocprim::other'
    return name[:60]


def main():
    root, out = sys.argv[1], sys.argv[2]
    lines = ['# rocprofv3 summary: %s' % os.path.basename(os.path.normpath(root)), '']
    stats = glob.glob(os.path.join(root, 'trace', '*', '*_kernel_stats.csv'))
    if stats:
        lines += ['## kernel trace (--kernel-trace --stats)', '',
                  '| kernel | calls | avg ms | min ms | max ms | % |', '|---|---|---|---|---|---|']
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                lines.append('| %s | %s | %.4f | %.4f | %.4f | %.2f |' % (
                    short(row['Name']), row['Calls'], float(row['AverageNs']) / 1e6, float(row['MinNs']) / 1e6,
                    float(row['MaxNs']) / 1e6, float(row['Percentage'])))
        lines.append('')
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, 'pmc_*', '*', '*_counter_collection.csv')):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                pmc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    if pmc:
        lines += ['## PMC counters (one --pmc pass per counter group), mean per launch', '',
                  '| kernel | counter | launches | mean |', '|---|---|---|---|']
        for k in sorted(pmc):
            if not k.startswith('gcn::spmm') and not k.startswith('gcn::gemm'):
                continue
            for c in sorted(pmc[k]):
                v = pmc[k][c]
                lines.append('| %s | %s | %d | %.6g |' % (k, c, len(v), sum(v) / len(v)))
        lines.append('')
        for k in sorted(pmc):
            if 'FETCH_SIZE' in pmc[k] and 'WRITE_SIZE' in pmc[k] and k.startswith('gcn::spmm'):
                fs = sum(pmc[k]['FETCH_SIZE']) / len(pmc[k]['FETCH_SIZE'])
                ws = sum(pmc[k]['WRITE_SIZE']) / len(pmc[k]['WRITE_SIZE'])
                lines.append('- %s: corrected HBM-side bytes per launch = (2*%.6g + %.6g) KiB = %.4g GB' % (
                    k, fs, ws, (2 * fs + ws) * 1024 / 1e9))
            if 'TCC_HIT_sum' in pmc[k] and 'TCC_MISS_sum' in pmc[k] and k.startswith('gcn::spmm'):
                h = sum(pmc[k]['TCC_HIT_sum']) / len(pmc[k]['TCC_HIT_sum'])
                m = sum(pmc[k]['TCC_MISS_sum']) / len(pmc[k]['TCC_MISS_sum'])
                lines.append('- %s: L2 hit rate = %.3f' % (k, h / (h + m)))
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
