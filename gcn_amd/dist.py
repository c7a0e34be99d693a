"""Row-partitioned propagation over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-process; this module is what lets the same product run on a graph
split over 2/4/8 MI355X.  Rank p owns a contiguous block of rows of A_hat (a CSR shard with
GLOBAL column indices) and the same rows of every dense matrix.  One product is
all-gather(B) followed by the local product, so every output row is accumulated in exactly
the order of the single-GPU run and the results are bit-identical.
"""
import torch
import torch.distributed as dist

from . import sparse


def row_block(n_rows, rank, world):
    """Contiguous equal split; n_rows must be padded by the caller to a multiple of world."""
    if n_rows % world != 0:
        raise ValueError('n_rows=%d is not a multiple of the world size %d' % (n_rows, world))
    per = n_rows // world
    return rank * per, (rank + 1) * per


class RowShardedCSR(object):
    """The rows [row_begin, row_end) of a CSR matrix, columns still global."""

    def __init__(self, n_rows_global, n_cols, row_begin, row_end, local, rank, world):
        self.n_rows_global, self.n_cols = n_rows_global, n_cols
        self.row_begin, self.row_end = row_begin, row_end
        self.local = local
        self.rank, self.world = rank, world

    @classmethod
    def from_full(cls, n_rows, n_cols, indptr, indices, values, rank, world):
        r0, r1 = row_block(n_rows, rank, world)
        k0 = int(indptr[r0].item())
        k1 = int(indptr[r1].item())
        local_indptr = (indptr[r0:r1 + 1] - k0).contiguous()
        local = sparse.DeviceCSR(r1 - r0, n_cols, local_indptr, indices[k0:k1].clone(), values[k0:k1].clone())
        return cls(n_rows, n_cols, r0, r1, local, rank, world)


class AllGatherSpmm(object):
    """C_local = A_local . all_gather(B_local), buffers allocated once."""

    def __init__(self, shard, f, group=None, spmm=None):
        self.shard, self.f, self.group = shard, f, group
        self._spmm = spmm or sparse.spmm
        dev = shard.local.device
        self.b_full = torch.empty((shard.n_cols, f), dtype=torch.float32, device=dev)
        self.c_local = torch.empty((shard.local.n_rows, f), dtype=torch.float32, device=dev)

    def __call__(self, b_local, relu=False):
        if self.shard.world > 1:
            dist.all_gather_into_tensor(self.b_full, b_local.contiguous(), group=self.group)
        else:
            self.b_full.copy_(b_local)
        return self._spmm(self.shard.local, self.b_full, relu=relu, out=self.c_local)
