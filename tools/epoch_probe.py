"""A few training steps of the 2-layer GCN (256 -> 256 -> 16, dropout 0.5, one labelled node in 64) on the R-MAT
bench graph, for rocprofv3:  python tools/epoch_probe.py [--form reference|restructured] [--reorder hubs]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import sparse, dist as gdist            # noqa: E402
from gcn_amd.dist_train import ShardedGCN            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=24)
ap.add_argument('--form', default='reference')
ap.add_argument('--reorder', default=None)
ap.add_argument('--steps', type=int, default=4)
args = ap.parse_args()
n, f, hidden, classes = 1 << args.scale, 256, 256, 16
indptr, indices = sparse.rmat_csr(args.scale, 16, seed=42, self_loops=True)
d = sparse.degree_scale(sparse.row_sums(indptr, None), 'sym')
vals = sparse.normalize_values(indptr, indices, None, d, 'sym', want64=False)[1]
a = sparse.DeviceCSR(n, n, indptr, indices, vals)
shard = gdist.RowShardedCSR(n, n, 0, n, a, 0, 1)
x = sparse.synthetic_dense(n, f, seed=7)
rows = torch.arange(n, device='cuda')
labels = torch.nn.functional.one_hot((rows * 2654435761 >> 7) % classes, classes).to(torch.int32)
mask = (rows % 64 == 0).to(torch.int32)
form = (dict(aggregate_first=False, receptive_field=False, labelled_logits_only='backward') if args.form == 'reference'
        else dict(aggregate_first=False, receptive_field=False, labelled_logits_only=False, packed_operands=False)
        if args.form == 'dense-reference' else {})
model = ShardedGCN(shard, f, [hidden], classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.5, seed=123,
                   reorder=args.reorder, **form)
for _ in range(args.steps):
    loss = model.train_step(x, labels, mask)
torch.cuda.synchronize()
print('loss', float(loss))
