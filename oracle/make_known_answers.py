#!/usr/bin/env python3
"""Known-answer vectors of the north-star product, committed as tests/golden/spmm_known_answers.npz.

TEST INFRASTRUCTURE.  For each citation graph: A_hat (the float32 cast of the values the
reference's ``preprocess_adj`` returned, read from the committed fixture of that graph) times a
seeded dense matrix B, computed by scipy's float32 CSR product -- the serial nnz loop that
tf.sparse_tensor_dense_matmul runs on the CPU (gcn/layers.py:31, 202), each output row summed in
ascending column order with a separate multiply and add.  Widths 16 (the hidden layer) and the number
of classes of each dataset (PubMed: the classes only, to keep the file small).

B is not stored: it is ``gcn_oracle.synthetic_dense(n, f, seed)`` (integer hash, reproducible
everywhere).  Usage:  python oracle/make_known_answers.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gcn_oracle  # noqa: E402

CASES = (('cora', (16, 7)), ('citeseer', (16, 6)), ('pubmed_graph', (3,)))
SEED = 29


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    args = ap.parse_args()
    out = {'seed': np.int64(SEED)}
    for name, widths in CASES:
        fx = np.load(os.path.join(args.out, name + '.npz'))
        n = int(fx['n'])
        ip, ix, v32, _ = gcn_oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
        a32 = sp.csr_matrix((v32, ix, ip), shape=(n, n))
        for f in widths:
            b = gcn_oracle.synthetic_dense(n, f, seed=SEED)
            out['%s_f%d_f32' % (name, f)] = np.asarray(a32 @ b, dtype=np.float32)
    path = os.path.join(args.out, 'spmm_known_answers.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
