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


def _through_host(t, group):
    """gloo moves host memory; device tensors are staged explicitly so that the rehearsal of the
    multi-rank path on one GPU never depends on gloo's own device-copy streams.  RCCL ('nccl')
    works on device memory directly."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_to_all_rows(out, inp, out_splits=None, in_splits=None, group=None):
    if _through_host(out, group):
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(host_out)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    return out


def all_reduce_sum(t, group=None):
    if _through_host(t, group):
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_gather_rows(out, inp, group=None):
    if _through_host(out, group):
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    return out


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


def rmat_shard(scale, edge_factor, rank, world, seed=42, group=None, device='cuda'):
    """This rank's rows of A_hat = D^-1/2 (A + I) D^-1/2 of the R-MAT graph, generated without ever
    holding the other rows: the counter-based generator emits only the keys of the rank's row block
    (``sparse.rmat_csr_rows``), the ranks all-gather their degree vectors (8 bytes per vertex) and every
    rank scales its own entries.  Equal bit for bit to ``RowShardedCSR.from_full`` of the whole graph."""
    n = 1 << scale
    r0, r1 = row_block(n, rank, world)
    indptr, indices = sparse.rmat_csr_rows(scale, r0, r1, edge_factor, seed, self_loops=True, device=device)
    deg_local = (indptr[1:] - indptr[:-1]).to(torch.float64)          # row sums of A + I: every value is 1
    if world > 1:
        deg = torch.empty(n, dtype=torch.float64, device=deg_local.device)
        all_gather_rows(deg, deg_local.contiguous(), group=group)
    else:
        deg = deg_local
    d = sparse.degree_scale(deg, 'sym')
    del deg, deg_local
    # the scaling kernel looks d up by row number: give it the row pointers of the whole matrix, in
    # which the rows of the other ranks are empty
    nnz = int(indices.numel())
    whole = torch.empty(n + 1, dtype=torch.int64, device=indptr.device)
    whole[:r0] = 0
    whole[r0:r1 + 1] = indptr
    whole[r1 + 1:] = nnz
    vals = sparse.normalize_values(whole, indices, None, d, 'sym', want64=False)[1]
    del whole, d
    local = sparse.DeviceCSR(r1 - r0, n, indptr, indices, vals)
    return RowShardedCSR(n, n, r0, r1, local, rank, world)


def shard_with_columns(shard, col_keep):
    """The shard restricted to the columns flagged in ``col_keep`` (bool, one flag per global column),
    entry order inside every row unchanged.  When the unflagged rows of the dense operand are zero --
    the gradient of the logits of a semi-supervised model is zero outside the labelled nodes -- a
    product with this matrix adds the same nonzero terms in the same order as the product with the
    whole shard: the same bits on every row that product sums serially (hub rows, which it sums in
    chunks, agree to rounding)."""
    local = shard.local
    keep = col_keep[local.indices.to(torch.int64)]
    before = torch.zeros(local.nnz + 1, dtype=torch.int64, device=keep.device)
    torch.cumsum(keep, dim=0, out=before[1:])
    kept = type(local)(local.n_rows, local.n_cols, before[local.indptr].contiguous(), local.indices[keep].contiguous(),
                       local.values[keep].contiguous())
    return RowShardedCSR(shard.n_rows_global, shard.n_cols, shard.row_begin, shard.row_end, kept, shard.rank,
                         shard.world)


def shard_with_rows(shard, row_keep):
    """The shard with the entries of unflagged local rows removed (those rows become empty; ``row_keep``:
    bool, one flag per local row).  The product with it gives the flagged rows of the full product, bit
    for bit, and zeros elsewhere: the rows of the logits a masked loss reads (gcn/metrics.py:85-92)."""
    local = shard.local
    counts = local.indptr[1:] - local.indptr[:-1]
    kept_counts = counts * row_keep.to(counts.dtype)
    keep = torch.repeat_interleave(row_keep, counts)
    indptr = torch.zeros(local.n_rows + 1, dtype=torch.int64, device=counts.device)
    torch.cumsum(kept_counts, dim=0, out=indptr[1:])
    kept = type(local)(local.n_rows, local.n_cols, indptr, local.indices[keep].contiguous(),
                       local.values[keep].contiguous())
    return RowShardedCSR(shard.n_rows_global, shard.n_cols, shard.row_begin, shard.row_end, kept, shard.rank,
                         shard.world)


def shard_rows_compacted(shard, rows):
    """The rows ``rows`` (ascending local row numbers, int64) of the shard as a matrix of their own:
    row i of the result is local row rows[i], columns and entry order unchanged."""
    local = shard.local
    counts = (local.indptr[1:] - local.indptr[:-1])
    flag = torch.zeros(local.n_rows, dtype=torch.bool, device=rows.device)
    flag[rows] = True
    keep = torch.repeat_interleave(flag, counts)
    indptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(counts[rows], dim=0, out=indptr[1:])
    kept = type(local)(int(rows.numel()), local.n_cols, indptr, local.indices[keep].contiguous(),
                       local.values[keep].contiguous())
    return RowShardedCSR(shard.n_rows_global, shard.n_cols, shard.row_begin, shard.row_end, kept, shard.rank,
                         shard.world)


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
            all_gather_rows(self.b_full, b_local.contiguous(), group=self.group)
        else:
            self.b_full.copy_(b_local)
        return self._spmm(self.shard.local, self.b_full, relu=relu, out=self.c_local)


class HaloPlan(object):
    """Which rows of a row-sharded dense matrix every rank needs from every other rank.

    Set-up (integer work, once per graph): the distinct columns of the local shard that belong to OTHER
    ranks are sorted -- which groups them by owner rank, since owners hold contiguous blocks -- and the ranks
    exchange their request lists.  The operand of the local product then has two parts: the rank's own
    rows, read where they lie, and the received rows, in that sorted order; the shard's column indices are
    renumbered to positions in it (own rows first; the order of the entries inside a row does not change,
    so sums are bit-identical).  ``exchange(b_local)`` gathers the rows the others requested and moves them
    with one ``all_to_all_single``; the own rows are never packed or copied.  On a skewed graph most
    vertices have few neighbours, so a shard references a fraction of the columns (72 / 52 / 37 % at 2 / 4
    / 8 ranks on the scale-24 R-MAT), of which only the other ranks' (22 / 27 / 24 % of N) cross a link."""

    # own_first='auto': the own rows are multiplied first when they hold at least this share of the shard's entries
    OWN_FIRST_SHARE = 0.2

    def __init__(self, shard, group=None, csr_factory=None, operand_rows=None, own_first=False):
        """``operand_rows`` (ascending local row numbers, int64): the dense operand every rank passes to
        ``exchange`` holds only these rows of its block, in this order (every row of the block that anyone
        references must be among them).

        ``own_first``: cut the shard by column into the entries that read the rank's own rows and the entries
        that read received rows.  A product then multiplies the first part while the exchange is on the links
        and adds the second, range by range, as the rows arrive (``halo_product``): the work that needs no
        exchange hides the exchange.  A row's own-block entries are then summed before its other entries, not
        in column order: deterministic, equal to the single-GPU product to rounding (1e-5 of the largest
        element), no longer bit for bit.  False keeps the single-GPU order and bits.  'auto': True when the own
        part holds at least ``OWN_FIRST_SHARE`` of the shard's entries (2 and 4 ranks of an evenly spread
        graph), decided per rank from its own counts."""
        self.shard, self.group = shard, group
        self.own_first, self.own, self.remote = False, None, None
        local = shard.local
        dev = local.device
        world = shard.world
        per = shard.n_cols // world
        if shard.n_cols % world != 0:
            raise ValueError('n_cols=%d is not a multiple of the world size %d' % (shard.n_cols, world))
        self._collective = world > 1 or (dist.is_initialized() and dist.get_world_size(group) == 1)
        self._buffers, self._side = {}, None                         # stream of exchange_ranges: made on first use
        if not self._collective and operand_rows is None:
            # one process, no communicator: the operand holds every column's row, B is the whole operand
            self.compact, self.n_first, self.n_remote, self._identity = local, shard.n_cols, 0, True
            self.remote_rows = torch.zeros(0, dtype=torch.int64, device=dev)
            self.recv_splits = self.send_splits = [0]
            return
        r0 = shard.rank * per
        cols = local.indices.to(torch.int64)
        mine = (cols >= r0) & (cols < r0 + per)
        remote = torch.unique(cols[~mine])                           # sorted: grouped by owner
        # own rows: where they lie in the operand the caller passes (its whole block, or operand_rows of it)
        if operand_rows is not None:
            position = torch.full((per,), -1, dtype=torch.int64, device=dev)
            position[operand_rows] = torch.arange(operand_rows.numel(), dtype=torch.int64, device=dev)
            self.n_first = int(operand_rows.numel())
            own_cols = position[(cols - r0).clamp_(0, per - 1)]
            if bool((own_cols[mine] < 0).any()):
                raise ValueError('a referenced row of the own block is not among operand_rows')
        else:
            position = None
            self.n_first = per
            own_cols = cols - r0
        new_cols = torch.where(mine, own_cols, torch.searchsorted(remote, cols) + self.n_first).to(torch.int32)
        del own_cols, cols, mine
        owner = torch.div(remote, per, rounding_mode='floor')
        self.recv_counts = torch.bincount(owner, minlength=world).to(torch.int64)
        request = (remote - owner * per).contiguous()                # row offsets inside the owner's block
        # tell every owner which of its rows this rank wants
        self.send_counts = torch.empty(world, dtype=torch.int64, device=dev)
        # a 1-rank process group still goes through the collectives (same code path as N ranks)
        if self._collective:
            all_to_all_rows(self.send_counts, self.recv_counts, group=group)
        else:
            self.send_counts.copy_(self.recv_counts)
        self.recv_splits = [int(v) for v in self.recv_counts.tolist()]
        self.send_splits = [int(v) for v in self.send_counts.tolist()]
        self.send_rows = torch.empty(sum(self.send_splits), dtype=torch.int64, device=dev)
        if self._collective:
            all_to_all_rows(self.send_rows, request, self.send_splits, self.recv_splits, group=group)
        else:
            self.send_rows.copy_(request)
        if position is not None:                                     # row numbers -> positions in the operand
            self.send_rows = position[self.send_rows]
            if self.send_rows.numel() and int(self.send_rows.min()) < 0:
                raise ValueError('a requested row is not among operand_rows')
        self.n_remote = int(remote.numel())
        self.remote_rows = remote.contiguous()       # global row numbers of the received rows, in operand order
        make = csr_factory or sparse.DeviceCSR
        self._identity = False
        in_own = new_cols < self.n_first
        before = torch.zeros(local.nnz + 1, dtype=torch.int64, device=dev)
        torch.cumsum(in_own, dim=0, out=before[1:])
        if own_first == 'auto':
            own_first = world > 1 and local.nnz > 0 and int(before[-1]) >= self.OWN_FIRST_SHARE * local.nnz
        self.own_first = bool(own_first)
        if not self.own_first:
            del in_own, before
            self.compact = make(local.n_rows, self.n_first + self.n_remote, local.indptr, new_cols, local.values)
            return
        # two matrices over the same rows; the order of the entries inside a row is kept in both
        own_indptr = before[local.indptr].contiguous()
        del before
        self.own = make(local.n_rows, self.n_first, own_indptr, new_cols[in_own].contiguous(),
                        local.values[in_own].contiguous())
        in_own.logical_not_()
        self.remote = make(local.n_rows, self.n_remote, local.indptr - own_indptr,
                           (new_cols[in_own] - self.n_first).contiguous(), local.values[in_own].contiguous())
        self.compact = None

    def multiply(self, b_local, received, relu=False, out=None, spmm=None, **extra):
        """``compact . [b_local; received]`` without stacking the two parts (``sparse.spmm(first=...)``); a
        multiply function of the caller's (tests inject the CPU oracle) gets the stacked operand."""
        if self._identity:
            return (spmm or sparse.spmm)(self.compact, b_local, relu=relu, out=out, **extra)
        if b_local.shape[0] != self.n_first:
            raise ValueError('the operand has %d rows, the plan expects %d' % (b_local.shape[0], self.n_first))
        if self.own_first:
            out = self.multiply_own(b_local, out=out, spmm=spmm, wide_split=extra.get('wide_split', False))
            return self.add_remote(received, out, relu=relu, spmm=spmm, **extra)
        if spmm is None or spmm is sparse.spmm:
            return sparse.spmm(self.compact, received, relu=relu, out=out, first=b_local, **extra)
        return spmm(self.compact, torch.cat([b_local, received]), relu=relu, out=out, **extra)

    def multiply_own(self, b_local, out=None, spmm=None, wide_split=False):
        """First half of an own-first product: ``own . b_local`` (no activation: the sum is not complete)."""
        if spmm is None or spmm is sparse.spmm:
            return sparse.spmm(self.own, b_local, out=out, wide_split=wide_split)
        return spmm(self.own, b_local, out=out)

    def add_remote(self, received, partial, relu=False, spmm=None, **extra):
        """Second half: ``partial <- act(remote . received + partial)`` in place, with the store dropout of
        ``extra['out_dropout']`` if given.  ``partial`` may be a column range of the first half's output."""
        if spmm is None or spmm is sparse.spmm:
            return sparse.spmm_add(self.remote, received, partial, relu=relu, out=partial, **extra)
        # a multiply function of the caller's (the CPU oracle in tests) takes the addend as ``add=``
        return spmm(self.remote, received, relu=relu, out=partial, add=partial, **extra)

    @property
    def matrices(self):
        """The CSR matrices a product of this plan multiplies (one, or the own and the remote part)."""
        return (self.own, self.remote) if self.own_first else (self.compact,)

    def drop_buffers(self, keep_widths=()):
        """Free the cached send / receive buffers of every width not in ``keep_widths``."""
        for key in [k for k in self._buffers if k[0] not in keep_widths]:
            del self._buffers[key]

    def _buffers_for(self, f, slot, device):
        bufs = self._buffers.get((f, slot))
        if bufs is None:
            bufs = (torch.empty((self.send_rows.numel(), f), dtype=torch.float32, device=device),
                    torch.empty((self.n_remote, f), dtype=torch.float32, device=device))
            self._buffers[(f, slot)] = bufs
        return bufs

    def exchange_ranges(self, b_local, ranges):
        """The exchange of every column range of ``ranges``, all issued now: ``[(rows, handle), ...]`` in
        range order, ``handle.wait()`` ordering the current stream after that range's arrival.

        On the GPU nothing of this runs on the caller's stream.  The send buffers are packed on a side
        stream of the plan's own -- which first waits for what the caller has enqueued so far, the
        producer of ``b_local`` and the last readers of the buffers among it -- and every range's
        collective is handed to RCCL from that stream, so RCCL's stream waits for the packing of its own
        range and for nothing else.  Range i + 1 is therefore packed while range i is on the links, all
        the ranges follow each other on the links without a gap, and the caller's products meet neither the
        packing nor a collective that had to wait for an earlier product.  Every range has buffers of its
        own (together: one full-width pair)."""
        if self._identity:
            return [(None, _Done) for lo, hi in ranges]
        staged = self._collective and _through_host(b_local, self.group)
        if not b_local.is_cuda or staged:
            # host tensors, and the gloo rehearsal that stages device tensors through the host: in order
            return [(self.exchange(b_local[:, lo:hi], slot=i), _Done) for i, (lo, hi) in enumerate(ranges)]
        dev = b_local.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        entry = torch.cuda.Event()
        entry.record(torch.cuda.current_stream(dev))
        self._side.wait_event(entry)
        out = []
        with torch.cuda.stream(self._side):
            for i, (lo, hi) in enumerate(ranges):
                send_buf, b_halo = self._buffers_for(hi - lo, i, dev)
                sparse.gather_rows(b_local[:, lo:hi], self.send_rows, send_buf)
                if self._collective:
                    work = dist.all_to_all_single(b_halo, send_buf, self.recv_splits, self.send_splits,
                                                  group=self.group, async_op=True)
                else:                                   # one process standing in for a rank: a device copy
                    b_halo.copy_(send_buf)
                    done = torch.cuda.Event()
                    done.record(self._side)
                    work = _AfterEvent(done)
                out.append((b_halo, work))
        return out

    def exchange(self, b_local, slot=0):
        """The rows of the other ranks' blocks this rank's shard references, [n_remote x f], grouped by owner,
        moved on the caller's stream (the exchange a product of one column range waits for as a whole).
        ``slot`` picks one of the cached buffer pairs of this width."""
        if self._identity:                      # one process: nothing to receive
            return None
        send_buf, b_halo = self._buffers_for(b_local.shape[1], slot, b_local.device)
        if b_local.is_cuda:
            sparse.gather_rows(b_local, self.send_rows, send_buf)      # column slices are read in place
        else:
            torch.index_select(b_local, 0, self.send_rows, out=send_buf)
        if self._collective:
            all_to_all_rows(b_halo, send_buf, self.recv_splits, self.send_splits, group=self.group)
        else:
            b_halo.copy_(send_buf)
        return b_halo


class _Done(object):
    """Stands for the handle of an exchange that has already completed."""
    @staticmethod
    def wait():
        return True


class _AfterEvent(object):
    """Handle of work enqueued on another stream: ``wait()`` orders the current stream after it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


def column_chunks(f, chunks, progressive=True):
    """Split f feature columns into at most ``chunks`` contiguous ranges whose widths are multiples
    of 4 floats (128-bit loads stay possible in every range).

    ``progressive``: widths in the proportions 1 : 1 : 2 : 4 : ... (32, 32, 64, 128 for f = 256,
    chunks = 4).  The exchange of the first range is the only one a pipelined product waits for, so it
    should be short; each later exchange hides behind the product of the range before it as long as
    a range is at most about twice that one -- moving a column is faster than multiplying it on 2, 4
    and 8 GPUs of the scale-24 graph."""
    if chunks <= 1 or f < 8:
        return [(0, f)]
    quads = (f + 3) // 4
    chunks = min(chunks, quads)
    if progressive:
        while chunks > 1 and quads < (1 << (chunks - 1)):
            chunks -= 1
        units = 1 << (chunks - 1)
        shares = [1] + [1 << max(i - 1, 0) for i in range(1, chunks)]      # 1, 1, 2, 4, ...
        edges, acc = [0], 0
        for share in shares:
            acc += share
            edges.append(min(f, 4 * ((quads * acc) // units)))
    else:
        edges = [0] + [min(f, 4 * ((quads * (i + 1)) // chunks)) for i in range(chunks)]
    edges[-1] = f
    return [(lo, hi) for lo, hi in zip(edges, edges[1:]) if hi > lo]


class HaloSpmm(object):
    """C_local = A_local . B where every rank receives only the rows of B its shard references.

    ``own_first`` (see HaloPlan): the entries that read the rank's own rows are multiplied while the exchange
    is on the links; off by default, which keeps the bits of the single-GPU product.

    With ``chunks`` > 1 the feature columns are cut into that many ranges whose exchanges are all issued
    at once and follow each other on the links while the products of the ranges that have arrived run
    (each output column depends on the same column of B only, and every row still accumulates in
    storage order, so the result does not change by a bit).  The first range's exchange is the only one
    the product waits for."""

    def __init__(self, shard, f, group=None, spmm=None, csr_factory=None, plan=None, chunks=1, own_first=False):
        self.shard, self.f = shard, f
        self._spmm = spmm or sparse.spmm
        self.plan = plan or HaloPlan(shard, group=group, csr_factory=csr_factory, own_first=own_first)
        self.compact = self.plan.compact
        self.n_remote = self.plan.n_remote
        self.recv_splits = self.plan.recv_splits
        self.ranges = column_chunks(f, chunks)
        self.c_local = torch.empty((shard.local.n_rows, f), dtype=torch.float32, device=shard.local.device)

    def __call__(self, b_local, relu=False):
        return halo_product(self.plan, b_local, self.c_local, relu, self.ranges, self._spmm)

    # width shares of the candidate pipelines: how long the first (exposed) exchange is against how
    # much the narrower products cost depends on the ratio of link to memory speed, which differs
    # between 2, 4 and 8 ranks -- so it is measured, not assumed
    # (the equal splits in four and eight win where the links are slow: the products then wait for rows anyway)
    CANDIDATES = ((1,), (1, 7), (1, 3), (1, 1), (1, 1, 6), (1, 2, 5), (1, 1, 2), (1, 1, 2, 4), (2, 3, 4, 7),
                  (1, 1, 1, 1), (1, 1, 1, 1, 1, 1, 1, 1))

    def autotune(self, b_local, reps=2, clock=None):
        """Time every candidate split of the feature columns on this communicator (max over the ranks,
        so that every rank picks the same one) and keep the fastest.  Returns [(ranges, ms), ...]."""
        import time
        world = self.shard.world
        clock = clock or time.perf_counter
        seen, results = set(), []
        for shares in self.CANDIDATES:
            ranges = column_shares(self.f, shares)
            if tuple(ranges) in seen:
                continue
            seen.add(tuple(ranges))
            self.ranges = ranges
            self(b_local)                                            # buffers of these widths, untimed
            if world > 1:
                dist.barrier(group=self.plan.group)
            if b_local.is_cuda:
                torch.cuda.synchronize()
            t0 = clock()
            for _ in range(reps):
                self(b_local)
            if b_local.is_cuda:
                torch.cuda.synchronize()
            ms = torch.tensor([(clock() - t0) / reps * 1e3], dtype=torch.float64)
            if world > 1:
                ms = ms.to(b_local.device) if dist.get_backend(self.plan.group) != 'gloo' else ms
                dist.all_reduce(ms, op=dist.ReduceOp.MAX, group=self.plan.group)
            results.append((ranges, float(ms.item())))
        self.ranges = min(results, key=lambda r: r[1])[0]
        self.plan.drop_buffers(keep_widths={hi - lo for lo, hi in self.ranges})
        return results


def column_shares(f, shares):
    """Contiguous column ranges of f columns with widths in the proportions ``shares`` (multiples of 4)."""
    quads = (f + 3) // 4
    total = sum(shares)
    edges, acc = [0], 0
    for share in shares:
        acc += share
        edges.append(min(f, 4 * ((quads * acc) // total)))
    edges[-1] = f
    return [(lo, hi) for lo, hi in zip(edges, edges[1:]) if hi > lo]


def halo_product(plan, b_local, out, relu=False, ranges=None, spmm=None, out_dropout=None):
    """out = A_local . B over column ranges: every range's exchange is issued at once (HaloPlan.exchange_ranges)
    and the product of a range starts when its rows have arrived, so only the first range's exchange is exposed.
    ``out_dropout`` (sparse.OutDropout over the full width, index_offset = first local row * width):
    the output is stored already dropped."""
    f = b_local.shape[1]
    ranges = ranges or [(0, f)]
    extra = {}
    if out_dropout is not None:
        out_dropout = sparse.OutDropout(out_dropout.keep_prob, out_dropout.seed, out_dropout.stream_id,
                                        out_dropout.index_offset, f)
    if plan.own_first:
        # the exchange of every range goes out, the entries that read own rows are multiplied meanwhile, then
        # each range's received rows are multiplied and added to what is there
        handles = plan.exchange_ranges(b_local, ranges)
        plan.multiply_own(b_local, out=out, spmm=spmm, wide_split=False)
        for (c0, c1), (rows, work) in zip(ranges, handles):
            work.wait()
            if out_dropout is not None:
                extra['out_dropout'] = out_dropout.columns(c0)
            plan.add_remote(rows, out[:, c0:c1], relu=relu, spmm=spmm, wide_split=f >= 128, **extra)
        return out
    if len(ranges) == 1:
        if out_dropout is not None:
            extra['out_dropout'] = out_dropout
        return plan.multiply(b_local, plan.exchange(b_local), relu=relu, out=out, spmm=spmm, **extra)
    for (c0, c1), (rows, work) in zip(ranges, plan.exchange_ranges(b_local, ranges)):
        work.wait()
        if out_dropout is not None:
            extra['out_dropout'] = out_dropout.columns(c0)
        plan.multiply(b_local[:, c0:c1], rows, relu=relu, out=out[:, c0:c1], spmm=spmm, wide_split=f >= 128, **extra)
    return out
