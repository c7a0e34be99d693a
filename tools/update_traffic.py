#!/usr/bin/env python3
"""Write profiles/traffic.json from the counter passes of tools/profile_passes.sh on the bench product.

    python tools/update_traffic.py <profile dir> <committed summary path> [--scale 24 --edge-factor 16 --features 256]

The record holds the bytes one launch of the row kernel moves behind the L2 -- read requests of the L2 to
the fabric (all of 128 bytes on this kernel, counted by size) plus WRITE_SIZE -- and the sha256 of
gcn_amd/csrc/spmm.hip at the time, so that bench.py quotes it only for the kernels it was measured on.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument('profile_dir')
ap.add_argument('summary_path')
ap.add_argument('--scale', type=int, default=24)
ap.add_argument('--edge-factor', type=int, default=16)
ap.add_argument('--features', type=int, default=256)
args = ap.parse_args()

vals = collections.defaultdict(list)
for path in glob.glob(os.path.join(args.profile_dir, 'pmc_*', '*', '*_counter_collection.csv')):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if 'spmm_rows_kernel' in row['Kernel_Name']:
                vals[row['Counter_Name']].append(float(row['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
reads = 128 * mean['TCC_EA0_RDREQ_128B_sum'] + 64 * mean['TCC_EA0_RDREQ_64B_sum'] + 32 * mean['TCC_EA0_RDREQ_32B_sum']
writes = 1024 * mean['WRITE_SIZE']
with open(os.path.join(ROOT, 'gcn_amd', 'csrc', 'spmm.hip'), 'rb') as fh:
    digest = hashlib.sha256(fh.read()).hexdigest()
rec = {
    'kernel': 'spmm_rows_kernel', 'scale': args.scale, 'edge_factor': args.edge_factor, 'features': args.features,
    'read_requests_128b': mean['TCC_EA0_RDREQ_128B_sum'], 'read_requests_64b': mean['TCC_EA0_RDREQ_64B_sum'],
    'read_requests_32b': mean['TCC_EA0_RDREQ_32B_sum'], 'fetch_size_kib': mean['FETCH_SIZE'],
    'write_size_kib': mean['WRITE_SIZE'], 'tcc_hit': mean['TCC_HIT_sum'], 'tcc_miss': mean['TCC_MISS_sum'],
    'l2_miss_bytes_per_launch': reads + writes,
    'spmm_hip_sha256': digest,
    'source': '%s (rocprofv3 --pmc passes of tools/orderings.py; read requests by size x their bytes + WRITE_SIZE; '
              'Infinity-Cache hits are included: this is traffic behind the L2, an upper bound of the HBM bytes)'
              % args.summary_path,
}
with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
    json.dump([rec], fh, indent=1)
print(json.dumps(rec, indent=1))
