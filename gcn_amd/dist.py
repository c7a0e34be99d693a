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


class HaloSpmm(object):
    """C_local = A_local . B where every rank receives only the rows of B its shard references.

    Set-up (integer work, once per graph): the distinct columns of the local shard are sorted --
    which groups them by owner rank, since owners hold contiguous blocks -- the shard's column
    indices are renumbered to positions in that list (the order of the entries inside a row does
    not change, so sums are bit-identical), and the ranks exchange their request lists.  Per
    product: gather the requested local rows, one ``all_to_all_single``, local product.  On a
    skewed graph most vertices have few neighbours, so a shard references a fraction of the
    columns (about a quarter at 8 ranks on the scale-24 R-MAT) and the exchange moves that
    fraction of what an all-gather moves."""

    def __init__(self, shard, f, group=None, spmm=None, csr_factory=None):
        self.shard, self.f, self.group = shard, f, group
        self._spmm = spmm or sparse.spmm
        local = shard.local
        dev = local.device
        world, rank = shard.world, shard.rank
        per = shard.n_cols // world
        if shard.n_cols % world != 0:
            raise ValueError('n_cols=%d is not a multiple of the world size %d' % (shard.n_cols, world))
        cols = local.indices.to(torch.int64)
        needed = torch.unique(cols)                                  # sorted: grouped by owner
        new_cols = torch.searchsorted(needed, cols).to(torch.int32)
        owner = torch.div(needed, per, rounding_mode='floor')
        self.recv_counts = torch.bincount(owner, minlength=world).to(torch.int64)
        request = (needed - owner * per).contiguous()                # row offsets inside the owner's block
        # tell every owner which of its rows this rank wants
        self.send_counts = torch.empty(world, dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_to_all_single(self.send_counts, self.recv_counts, group=group)
        else:
            self.send_counts.copy_(self.recv_counts)
        self.recv_splits = [int(v) for v in self.recv_counts.tolist()]
        self.send_splits = [int(v) for v in self.send_counts.tolist()]
        self.send_rows = torch.empty(sum(self.send_splits), dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_to_all_single(self.send_rows, request, self.send_splits, self.recv_splits, group=group)
        else:
            self.send_rows.copy_(request)
        make = csr_factory or sparse.DeviceCSR
        self.compact = make(local.n_rows, int(needed.numel()), local.indptr, new_cols, local.values)
        self.n_needed = int(needed.numel())
        self.send_buf = torch.empty((self.send_rows.numel(), f), dtype=torch.float32, device=dev)
        self.b_halo = torch.empty((self.n_needed, f), dtype=torch.float32, device=dev)
        self.c_local = torch.empty((local.n_rows, f), dtype=torch.float32, device=dev)

    def __call__(self, b_local, relu=False):
        torch.index_select(b_local, 0, self.send_rows, out=self.send_buf)
        if self.shard.world > 1:
            dist.all_to_all_single(self.b_halo, self.send_buf, self.recv_splits, self.send_splits, group=self.group)
        else:
            self.b_halo.copy_(self.send_buf)
        return self._spmm(self.compact, self.b_halo, relu=relu, out=self.c_local)
