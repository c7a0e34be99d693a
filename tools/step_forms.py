#!/usr/bin/env python3
"""Loss trajectories of the three forms of the large-graph training step (reference order; first layer
aggregated first; receptive field of the labelled nodes) on a scale-20 R-MAT graph: 40 Adam steps with
dropout 0.5 from the same seed.  The forms differ by rounding only."""
import sys, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gcn_amd import sparse, dist as gdist
from gcn_amd.dist_train import ShardedGCN
scale, f_in, hidden, classes = 20, 256, 256, 16
n = 1 << scale
indptr, indices = sparse.rmat_csr(scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
shard = gdist.RowShardedCSR.from_full(n, n, indptr, indices, vals, 0, 1)
x = sparse.synthetic_dense(n, f_in, seed=7)
rows = torch.arange(n, device='cuda')
labels = torch.nn.functional.one_hot((rows * 2654435761 >> 7) % classes, classes).to(torch.int32)
mask = (rows % 64 == 0).to(torch.int32)
out = {}
for name, kw in (('reference order, all rows', dict(aggregate_first=False, receptive_field=False)),
                 ('aggregate first, all rows', dict(aggregate_first=True, receptive_field=False)),
                 ('aggregate first, receptive field', dict(aggregate_first=True, receptive_field=True))):
    m = ShardedGCN(shard, f_in, [hidden], classes, dropout=0.5, seed=123, **kw)
    out[name] = [float(m.train_step(x, labels, mask)) for _ in range(40)]
ref = np.array(out['reference order, all rows'])
for name, l in out.items():
    l = np.array(l)
    print('%-34s loss[0]=%.7f loss[39]=%.7f  max |rel diff to reference order| = %.2e' % (name, l[0], l[-1], np.abs(l / ref - 1).max()))
