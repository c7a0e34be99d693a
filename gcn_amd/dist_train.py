"""A GCN training step on a row-sharded graph (one process per GPU).

Forward per layer, on rank p:  P_p = H_p . W (local GEMM, W replicated)  ->  exchange of the rows of
P this rank's shard references  ->  Z_p = A_p . P (local product).  For a symmetric support the
backward product dP = A^T . dZ has the same shape: dP_p = A_p . exchange(dZ), so no scatter-add and
no change in any row's accumulation order.  Weight gradients are partial sums over the local rows
and are all-reduced; the masked loss is normalised by the global mask count.  This is the
multi-GPU form of the step the reference times at train.py:203-205 (forward + backward + Adam).
"""
import numpy as np
import torch

from . import graph, sparse
from .dist import (HaloPlan, all_gather_rows, all_reduce_sum, column_chunks, column_shares, halo_product,
                   shard_rows_compacted, shard_with_columns, shard_with_rows)
from .layers import _DroppedInputGemmFn, _GatedGemmFn, _GemmFn
from .models import TfAdam


def _product(plan, b_local, relu, chunks, out_dropout=None):
    """A_local . exchange(B) into a fresh tensor; wide operands pipeline the exchange over column ranges."""
    out = torch.empty((plan.shard.local.n_rows, b_local.shape[1]), dtype=torch.float32, device=b_local.device)
    f = b_local.shape[1]
    if f < 128:
        ranges = [(0, f)]
    elif isinstance(chunks, (tuple, list)):           # width shares, e.g. the ones HaloSpmm.autotune picked
        ranges = column_shares(f, chunks)
    else:
        ranges = column_chunks(f, chunks)
    return halo_product(plan, b_local, out, relu, ranges, out_dropout=out_dropout)


class _HaloSpmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, b_local, fuse_relu, chunks, gate_here, out_dropout=None, backward_plan=None,
                pack_grad=False):
        """``gate_here`` False: the backward of the ReLU (and of ``out_dropout``, the next layer's input
        dropout applied in this product's store) is applied by whoever consumes this output -- the
        gradient that arrives here is already the one of the pre-activation.
        ``backward_plan``: the plan of the backward product when the arriving gradient is known to be zero
        outside some rows (the support restricted to those columns: fewer entries, a smaller exchange)."""
        out = _product(plan, b_local, fuse_relu, chunks, out_dropout)
        ctx.plan, ctx.gate, ctx.chunks = backward_plan or plan, fuse_relu and gate_here, chunks
        # ``pack_grad``: the gradient that arrives is zero wherever the unit was inactive or dropped; on one GPU
        # and for 128 < f <= 256 the backward product gathers it in packed form (same sums, fewer lines per row)
        ctx.pack_grad = bool(pack_grad) and fuse_relu
        if ctx.gate:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        if ctx.gate:
            out, = ctx.saved_tensors
            grad = sparse.dropout(grad, 1.0, 0, 0, gate=out)
        plan = ctx.plan
        grad = grad.contiguous()
        # symmetric support: rows of A^T are rows of A
        if ctx.pack_grad and plan._identity and 128 < grad.shape[1] <= sparse.PackedRows.MAX_F:
            out = sparse.spmm_packed(plan.compact, sparse.pack_rows(grad))
        elif (ctx.pack_grad and not plan.own_first and 128 < grad.shape[1] <= sparse.PackedRows.MAX_F
              and grad.shape[0] == plan.n_first):
            # N ranks, single-GPU order of additions: own and received rows of the gradient packed into one
            # operand [own rows | received rows]; the same entries in the same order as the dense two-part product
            received = plan.exchange(grad)
            n_own, n_got = plan.n_first, plan.n_remote
            packed = getattr(plan, '_packed_gradient', None)
            if packed is None or packed.n_rows != n_own + n_got or packed.f != grad.shape[1]:
                packed = plan._packed_gradient = sparse.PackedRows.empty(n_own + n_got, grad.shape[1], grad.device)
            sparse.pack_rows(grad, out=packed.rows(0, n_own))
            if n_got:
                sparse.pack_rows(received, out=packed.rows(n_own, n_own + n_got))
            out = sparse.spmm_packed(plan.compact, packed)
        else:
            out = _product(plan, grad, False, ctx.chunks)
        return None, out, None, None, None, None, None, None


class _HiddenThenClassesFn(torch.autograd.Function):
    """One GPU, a hidden layer of 129..256 units followed by a layer of at most 16 outputs, dropout on:
    ``dropout(relu(A . P)) . W_next`` as one node of the tape, so that the backward can keep the gradient of the
    hidden layer packed from the moment it is formed: ``where(h > 0, (dP2 . W_next^T) / keep, 0)`` goes straight
    into packed rows (sparse.gemm_gated_pack; three quarters of it are zeros) and the backward product gathers
    those -- the dense gradient matrix, its write and the pass that packs it never happen.  Same forward as
    _HaloSpmmFn followed by _GatedGemmFn; the gradient agrees with theirs up to the order of the <= 16 additions
    of its small product."""

    @staticmethod
    def usable(plan, p, w_next, keep):
        return (plan._identity and keep < 1.0 and 128 < p.shape[1] <= sparse.PackedRows.MAX_F
                and w_next.shape[1] <= sparse.GATED_PACK_MAX_K)

    @staticmethod
    def forward(ctx, plan, p, w_next, keep, out_dropout, chunks, grad_rows=None, grad_support=None):
        """``grad_rows`` (ascending int64 row numbers, or None): the gradient that will arrive is zero in every
        other row (the backward product of the logits ran over the labelled nodes' columns: only their neighbours
        get a non-zero row), so the weight gradient sums over these rows only -- the other terms are zeros.
        ``grad_support``: the support restricted to the columns ``grad_rows`` (the entries that meet a non-zero
        row of this layer's gradient) for the backward product; None: the whole support."""
        h = _product(plan, p, True, chunks, out_dropout)
        ctx.plan, ctx.keep, ctx.grad_rows, ctx.grad_support = plan, keep, grad_rows, grad_support
        ctx.save_for_backward(h, w_next)
        return sparse.gemm(h, w_next)

    @staticmethod
    def backward(ctx, grad):
        h, w_next = ctx.saved_tensors
        grad = grad.contiguous()
        d_w = None
        rows = ctx.grad_rows
        if rows is not None and 2 * rows.numel() >= h.shape[0]:
            rows = None                      # most rows are live: no gain in listing them
        if ctx.needs_input_grad[2]:
            if rows is not None:
                h_rows = torch.empty((rows.numel(), h.shape[1]), dtype=torch.float32, device=h.device)
                g_rows = torch.empty((rows.numel(), grad.shape[1]), dtype=torch.float32, device=h.device)
                sparse.gather_rows(h, rows, h_rows)
                sparse.gather_rows(grad, rows, g_rows)
                d_w = sparse.gemm_tn(h_rows, g_rows)
            else:
                d_w = sparse.gemm_tn(h, grad)
        d_p = None
        if ctx.needs_input_grad[1]:
            # rows of the gradient outside ``rows`` are zero: they come out empty without their gate being read
            packed = sparse.gemm_gated_pack(grad, w_next, h, ctx.keep, rows=rows)
            # symmetric support: rows of A^T are rows of A; the entries that would gather an empty row are left out
            support = ctx.grad_support if (rows is not None and ctx.grad_support is not None) else ctx.plan.compact
            d_p = sparse.spmm_packed(support, packed)
        return None, d_p, d_w, None, None, None, None, None


class _ShardDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep_prob, seed, stream_id, index_offset):
        ctx.args = (keep_prob, seed, stream_id, index_offset)
        return sparse.dropout(x, keep_prob, seed, stream_id, index_offset=index_offset)

    @staticmethod
    def backward(ctx, grad):
        keep_prob, seed, stream_id, index_offset = ctx.args
        return sparse.dropout(grad, keep_prob, seed, stream_id, index_offset=index_offset), None, None, None, None


class _AggregatedFirstLayerFn(torch.autograd.Function):
    """act(S . W) for S = A . dropout(X), the first layer taken in the order (A . X') . W.

    A . (X' . W) and (A . X') . W are the same matrix up to rounding; when the input is not wider than
    the layer, aggregating first costs the same forward product and makes the backward one unnecessary:
    X has no gradient, and dW = S^T . dZ needs only S, which the forward pass already holds -- where the
    reference's order needs A^T . dZ, a second product as wide as the layer, before X'^T . (A^T . dZ)."""

    @staticmethod
    def forward(ctx, s, w, relu, out_dropout):
        # out_dropout: the next layer's input dropout, in this product's store; its backward is the
        # consumer's (_GatedGemmFn)
        z = sparse.gemm(s, w, relu=relu, out_dropout=out_dropout)
        ctx.gate = relu and out_dropout is None
        if ctx.gate:
            ctx.save_for_backward(s, z)
        else:
            ctx.save_for_backward(s)
        return z

    @staticmethod
    def backward(ctx, grad):
        if ctx.gate:
            s, z = ctx.saved_tensors
            grad = sparse.dropout(grad, 1.0, 0, 0, gate=z)
        else:
            s, = ctx.saved_tensors
        return None, sparse.gemm_tn(s, grad.contiguous()), None, None


class ShardedGCN(object):
    """Dense-feature GCN (connection 'c' * L) over a ``RowShardedCSR`` with a symmetric support.

    ``aggregate_first`` (default: when the input is not wider than the first layer): the first layer is
    computed as (A . X') . W0 instead of A . (X' . W0) -- the same matrix up to rounding -- which drops the
    backward product of that layer (see _AggregatedFirstLayerFn)."""

    def __init__(self, shard, f_in, layer_size, n_classes, learning_rate=0.02, weight_decay=5e-4, dropout=0.5,
                 seed=0, group=None, chunks=4, aggregate_first=None, receptive_field=True, labelled_logits_only=True,
                 reorder=None, own_first=False, packed_operands=True, cache_feature_halo=True):
        """``packed_operands``: False keeps every product on dense operands (see _aggregate_dropped_input).

        ``cache_feature_halo`` (N ranks, first layer aggregated first, packed operands): the rows of the feature
        matrix that a rank's shard reads from other ranks are fetched ONCE per features tensor and kept; every step
        drops own and kept rows locally (a row's decisions depend on its global number only), so the step exchanges
        no wide rows at all (see _aggregate_dropped_input).  The features are an input that does not change between
        steps, like the support (the reference feeds the same matrix every step, train.py:180-183).

        ``own_first`` (False, True or 'auto'): handed to every HaloPlan of the model (see dist.HaloPlan): the
        entries that read a rank's own rows are multiplied while the exchange is on the links.

        ``reorder`` ('degree', 'hubs' or ('hubs', K); one rank only): the model works on the graph with its
        vertices relabelled once (utils.vertex_order); features, labels and masks given in the dataset's order
        are permuted on their way in (once per tensor object) and ``forward`` returns its rows in the dataset's
        order.  Dropout decisions are then drawn by the new row numbers."""
        self._order = None
        if reorder is not None:
            if shard.world != 1:
                raise ValueError('reorder relabels the whole graph: relabel before sharding (one rank only here)')
            from .utils import relabelled_csr, vertex_order
            new_id, old_id = vertex_order(shard.local.indptr, reorder)
            shard = type(shard)(shard.n_rows_global, shard.n_cols, shard.row_begin, shard.row_end,
                                relabelled_csr(shard.local, new_id), shard.rank, shard.world)
            self._order = (new_id.to(torch.int64), old_id.to(torch.int64))
            self._moved = {}
        self.shard, self.group = shard, group
        # ranges of feature columns an exchange is pipelined over: a count, or a tuple of width shares
        self.chunks = chunks if shard.world > 1 else 1
        self.own_first = own_first
        # one GPU: dropout(X) goes to the first layer's product in packed form (see _aggregate_dropped_input)
        self.packed_operands, self._pack_buffer = bool(packed_operands), None
        self.cache_feature_halo, self._feature_halos, self._in_order_twins = bool(cache_feature_halo), {}, {}
        self.plan = HaloPlan(shard, group=group, own_first=own_first)
        self.dropout, self.wd, self.seed = float(dropout), float(weight_decay), int(seed)
        sizes = [f_in] + list(layer_size) + [n_classes]
        self.aggregate_first = (f_in <= sizes[1]) if aggregate_first is None else bool(aggregate_first)
        # a training step of a two-layer model forms only the hidden rows the labelled nodes read
        self.receptive_field = bool(receptive_field)
        # a training step computes the last layer's product for labelled rows only (the masked loss reads no
        # other logit) and its backward product over labelled columns only; False: all rows, as the reference does
        # 'backward': all logits forward, but the backward product of the logits still skips the columns of
        # unlabelled nodes, whose rows of the gradient are exactly zero (the same terms in the same order)
        self.labelled_logits_only = labelled_logits_only if labelled_logits_only == 'backward' else bool(labelled_logits_only)
        dev = shard.local.device
        g = torch.Generator(device=dev)
        g.manual_seed(self.seed)                       # every rank draws the same replicated weights
        self.weights = []
        for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
            r = float(np.sqrt(6.0 / (fan_in + fan_out)))
            w = (torch.rand((fan_in, fan_out), generator=g, device=dev, dtype=torch.float32) * 2 - 1) * r
            self.weights.append(w.requires_grad_(True))
        self.optimizer = TfAdam(self.weights, learning_rate)
        self._mask_cache = {}
        self._grad_plans = {}
        self.step_count = 0

    def _all_reduce(self, t):
        if self.shard.world > 1:
            all_reduce_sum(t, group=self.group)
        return t

    def _logit_plans(self, mask_local):
        """(forward, backward) plans of the last layer's products in a training step.

        Forward: the masked loss reads the logits of labelled nodes only, so only their rows of
        A . (H' . W) are computed (the other rows of the support are emptied; their logits come out as 0
        and weigh 0 in the loss).  Backward: A^T . d logits.  The gradient of the masked loss
        is zero at every node outside the mask (gcn/metrics.py:85-92), so only the columns of labelled
        nodes contribute: the support is restricted to them once per mask (1/64 of the entries for one
        labelled node in 64) and the exchange shrinks with it.  The same terms in the same order as the
        product with the whole support."""
        hit = self._grad_plans.get(id(mask_local))
        if hit is None or hit[0] is not mask_local:
            flags = (mask_local != 0).to(torch.int32)
            if self.shard.world > 1:
                everyone = torch.empty(self.shard.n_cols, dtype=torch.int32, device=flags.device)
                all_gather_rows(everyone, flags.contiguous(), group=self.group)
            else:
                everyone = flags
            restricted = shard_with_columns(self.shard, everyone != 0)
            backward = HaloPlan(restricted, group=self.group, own_first=self.own_first)
            # the rows of that product's output that can be non-zero: the neighbours of labelled nodes
            lengths = restricted.local.indptr[1:] - restricted.local.indptr[:-1]
            backward.live_rows = torch.nonzero(lengths > 0).reshape(-1).contiguous()
            # one rank: the support without the columns of the other rows, for the backward product of the layer
            # before (its gradient is zero in those rows); built once per mask, only where it leaves entries out
            # N ranks: the plan of that layer's backward product over the columns of live rows only -- the rows of
            # its gradient that are zero are neither exchanged nor multiplied
            backward.hidden_plan = None
            if self.shard.world > 1:
                live_everywhere = torch.empty(self.shard.n_cols, dtype=torch.int32, device=lengths.device)
                all_gather_rows(live_everywhere, (lengths > 0).to(torch.int32).contiguous(), group=self.group)
                # a gradient that will be gathered packed keeps the single-GPU order of additions: its exchange is
                # small (live rows only), there is little for own rows to hide
                packs = self.packed_operands and 128 < self.weights[-1].shape[0] <= sparse.PackedRows.MAX_F
                backward.hidden_plan = HaloPlan(shard_with_columns(self.shard, live_everywhere != 0), group=self.group,
                                                own_first=False if packs else self.own_first)
                del live_everywhere
            backward.live_support = None
            if self.shard.world == 1 and 2 * backward.live_rows.numel() < lengths.numel():
                live = torch.zeros(self.shard.n_cols, dtype=torch.bool, device=lengths.device)
                live[backward.live_rows] = True
                backward.live_support = shard_with_columns(self.shard, live).local
                del live
            if self.labelled_logits_only == 'backward':       # every logit is formed, as the reference forms them
                forward = self.plan
            else:
                forward = HaloPlan(shard_with_rows(self.shard, flags != 0), group=self.group, own_first=self.own_first)
            hit = (mask_local, (forward, backward))
            self._grad_plans[id(mask_local)] = hit
        return hit[1]

    def _receptive_plans(self, mask_local):
        """Plans of a training step of a two-layer model cut down to what the masked loss can see.

        The loss reads the logits of labelled nodes; those read the hidden rows of their neighbours (the
        set R, one node in eight on the scale-24 R-MAT with one label in 64) and nothing else, so the
        hidden layer is computed for R only: S = A[R, :] . X', H = act(S . W0), P = H' . W1 are matrices of
        |R| rows, the logits are A[labelled, R] . P -- one row per labelled node -- and the backward product
        is A[R, labelled] . d logits.
        Every number that reaches the loss or a weight gradient is the one the full computation produces
        (dropout decisions are drawn by global row); rows outside R are never formed."""
        hit = self._grad_plans.get(('receptive', id(mask_local)))
        if hit is None or hit[0] is not mask_local:
            dev = mask_local.device
            flags = (mask_local != 0)
            labelled_rows = shard_with_rows(self.shard, flags)
            seen = torch.zeros(self.shard.n_cols, dtype=torch.float32, device=dev)
            seen[labelled_rows.local.indices.to(torch.int64)] = 1.0
            self._all_reduce(seen)                                   # columns some rank's labelled rows reference
            r_local = torch.nonzero(seen[self.shard.row_begin:self.shard.row_end] > 0).reshape(-1)
            del seen
            if self.shard.world > 1:
                everyone = torch.empty(self.shard.n_cols, dtype=torch.int32, device=dev)
                all_gather_rows(everyone, flags.to(torch.int32).contiguous(), group=self.group)
                everyone = everyone != 0
            else:
                everyone = flags
            lab_rows = torch.nonzero(flags).reshape(-1)              # the logits are formed for these rows only
            plans = {
                'rows': r_local,
                'row_ids': (r_local + self.shard.row_begin).contiguous(),
                'labelled': lab_rows,
                # with the features' halo kept on the rank this product waits for no exchange: nothing to hide
                'wide': HaloPlan(shard_rows_compacted(self.shard, r_local), group=self.group,
                                 own_first=False if self._keeps_feature_halo(self.weights[0].shape[0]) else self.own_first),
                'logits': HaloPlan(shard_rows_compacted(self.shard, lab_rows), group=self.group, operand_rows=r_local,
                                   own_first=self.own_first),
                'back': HaloPlan(shard_rows_compacted(shard_with_columns(self.shard, everyone), r_local),
                                 group=self.group, operand_rows=lab_rows, own_first=self.own_first),
            }
            hit = (mask_local, plans)
            self._grad_plans[('receptive', id(mask_local))] = hit
        return hit[1]

    def _forward_receptive(self, x_local, plans):
        keep = 1.0 - self.dropout
        drop = keep < 1.0
        w0, w1 = self.weights
        h = x_local
        s = self._aggregate_dropped_input(plans['wide'], h, keep)
        next_drop = None
        if drop:
            next_drop = sparse.OutDropout(keep, self.seed, self.step_count * 64 + 1, 0, w0.shape[1], plans['row_ids'])
        h = _AggregatedFirstLayerFn.apply(s, w0, True, next_drop)
        p = _GatedGemmFn.apply(h, w1, keep) if drop else _GemmFn.apply(h, w1)
        return _HaloSpmmFn.apply(plans['logits'], p, False, self.chunks, True, None, plans['back'])

    def _aggregate_dropped_input(self, plan, x_local, keep):
        """S = A . dropout(X) for the first layer taken as (A . X') . W0.  The features take no gradient, so their
        dropout is a plain pass -- and on one GPU, for 128 < f <= 256, not even that: dropout(X) is written in
        packed form (its kept values only, sparse.dropout_pack) and the product gathers the packed rows, fewer lines
        per row than the dense ones (same decisions, same values, same sums)."""
        f = x_local.shape[1]
        offset = self.shard.row_begin * f
        if keep >= 1.0:
            return _product(plan, x_local, False, self.chunks)
        if self.packed_operands and plan._identity and 128 < f <= sparse.PackedRows.MAX_F:
            packed = sparse.dropout_pack(x_local, keep, self.seed, self.step_count * 64, index_offset=offset,
                                         out=self._packed_rows(x_local.shape[0], f, x_local.device))
            return sparse.spmm_packed(plan.compact, packed)
        if self._keeps_feature_halo(f) and not plan._identity:
            # N ranks: X does not change between steps, so the rows of X this shard reads from other ranks were
            # fetched once (_feature_halo); own and kept rows are dropped here -- a kept row draws the decisions of
            # its global row, as its owner draws them -- straight into one packed operand [own rows | kept rows],
            # which the product gathers in the single-GPU order of additions.  No exchange in the step.
            plan = self._in_order(plan)
            halo = self._feature_halo(plan, x_local)
            n_own, n_kept = x_local.shape[0], halo.shape[0]
            packed = self._packed_rows(n_own + n_kept, f, x_local.device)
            sparse.dropout_pack(x_local, keep, self.seed, self.step_count * 64, index_offset=offset,
                                out=packed.rows(0, n_own))
            if n_kept:
                sparse.dropout_pack(halo, keep, self.seed, self.step_count * 64, row_ids=plan.remote_rows,
                                    out=packed.rows(n_own, n_own + n_kept))
            return sparse.spmm_packed(plan.compact, packed)
        h = sparse.dropout(x_local, keep, self.seed, self.step_count * 64, index_offset=offset)
        return _product(plan, h, False, self.chunks)

    def _packed_rows(self, n_rows, f, device):
        """The packed operand buffer of the first layer, reused over steps."""
        hit = self._pack_buffer
        if hit is None or hit.n_rows != n_rows or hit.f != f or hit.data.device != device:
            hit = self._pack_buffer = sparse.PackedRows.empty(n_rows, f, device)
        return hit

    def _keeps_feature_halo(self, f):
        # (a one-rank communicator goes the same way as N ranks, with nothing to fetch)
        return (self.cache_feature_halo and self.packed_operands and self.dropout > 0.0
                and 128 < f <= sparse.PackedRows.MAX_F)

    def _in_order(self, plan):
        """``plan`` itself, or for an own-rows-first plan its twin that keeps the single-GPU order of additions (one
        matrix over [own rows | received rows]); built once, by every rank at the same point of its first step."""
        if not plan.own_first:
            return plan
        hit = self._in_order_twins.get(id(plan))
        if hit is None or hit[0] is not plan:
            hit = (plan, HaloPlan(plan.shard, group=self.group, own_first=False))
            self._in_order_twins[id(plan)] = hit
        return hit[1]

    def _feature_halo(self, plan, x_local):
        """The rows of X that ``plan``'s shard reads from other ranks, [n_remote x f] in operand order: one exchange
        per (plan, features tensor OBJECT), kept for every later step (a tensor changed in place keeps its copy)."""
        key = (id(plan), id(x_local))
        hit = self._feature_halos.get(key)
        if hit is None or hit[0] is not plan or hit[1] is not x_local:
            received = plan.exchange(x_local)
            hit = (plan, x_local, received.clone())
            plan.drop_buffers()                       # the exchange buffers of this width are not used again
            self._feature_halos[key] = hit
        return hit[2]

    def _in_model_order(self, t):
        """A per-node tensor given in the dataset's order, in the order the model works in."""
        if self._order is None or t is None:
            return t
        hit = self._moved.get(id(t))
        if hit is None or hit[0] is not t:
            hit = (t, t[self._order[1]].contiguous())
            self._moved[id(t)] = hit
        return hit[1]

    def forward(self, x_local, train, logit_plans=None):
        """Dropout on every layer's input (gcn/models.py:60): a pass over the features for the first
        layer; for the others it is applied in the store of the product that makes the input, and its
        backward (with the ReLU's) in the store of the product that makes the input's gradient.
        Rows in, rows out: the dataset's order."""
        if self._order is not None:
            return self._forward(self._in_model_order(x_local), train, logit_plans)[self._order[0]]
        return self._forward(x_local, train, logit_plans)

    def _forward(self, x_local, train, logit_plans=None):
        keep = 1.0 - self.dropout if train else 1.0
        drop = keep < 1.0
        h = x_local
        last = len(self.weights) - 1
        ahead = None                       # a layer's X . W already formed by the layer before it
        for li, w in enumerate(self.weights):
            if ahead is not None:          # the product of the classes layer (it is the last: no activation)
                p, ahead = ahead, None
                if li == last and logit_plans is not None:
                    h = _HaloSpmmFn.apply(logit_plans[0], p, False, self.chunks, not drop, None, logit_plans[1])
                else:
                    h = _HaloSpmmFn.apply(self.plan, p, li != last, self.chunks, not drop, None, None,
                                          self.packed_operands)
                continue
            next_drop = None
            if drop and li != last:
                next_drop = sparse.OutDropout(keep, self.seed, self.step_count * 64 + li + 1,
                                              self.shard.row_begin * w.shape[1])
            if li == 0 and self.aggregate_first:
                s = self._aggregate_dropped_input(self.plan, h, keep)
                h = _AggregatedFirstLayerFn.apply(s, w, li != last, next_drop)
                continue
            if drop and li == 0 and not h.requires_grad:
                # the features take no gradient: their dropout is drawn as bits and applied inside the
                # product X' . W0 and inside its weight gradient (the same decisions as a pass over X)
                bits = sparse.dropout_bits(h.shape[0], h.shape[1], keep, self.seed, self.step_count * 64,
                                           index_offset=self.shard.row_begin * h.shape[1], device=h.device)
                p = _DroppedInputGemmFn.apply(h, w, bits, keep)
            else:
                if drop and li == 0:
                    offset = self.shard.row_begin * h.shape[1]
                    h = _ShardDropoutFn.apply(h, keep, self.seed, self.step_count * 64, offset)
                p = _GatedGemmFn.apply(h, w, keep) if drop and li > 0 else _GemmFn.apply(h, w)
            if li == last and logit_plans is not None:     # a training step: labelled rows forward, labelled columns backward
                h = _HaloSpmmFn.apply(logit_plans[0], p, False, self.chunks, not drop, None, logit_plans[1])
            elif (self.packed_operands and li + 1 == last and p.requires_grad
                  and _HiddenThenClassesFn.usable(self.plan, p, self.weights[li + 1], keep)):
                ahead = _HiddenThenClassesFn.apply(self.plan, p, self.weights[li + 1], keep, next_drop, self.chunks,
                                                   getattr(logit_plans[1], 'live_rows', None)
                                                   if logit_plans is not None else None,
                                                   getattr(logit_plans[1], 'live_support', None)
                                                   if logit_plans is not None else None)
            else:
                # the layer that feeds the classes layer of a training step: its gradient is zero outside the
                # neighbours of labelled nodes (see _logit_plans)
                hidden_backward = (getattr(logit_plans[1], 'hidden_plan', None)
                                   if (logit_plans is not None and li + 1 == last and drop) else None)
                h = _HaloSpmmFn.apply(self.plan, p, li != last, self.chunks, not drop, next_drop, hidden_backward,
                                      self.packed_operands)
        return h

    def _local_mask_mean(self, mask_local):
        """The value that makes the loss kernel weigh a local row by mask_i / (global mask count): it
        divides by mask_mean and by the local row count, so pass mean_global * n_global / n_local.
        The global count is reduced once per mask tensor."""
        hit = self._mask_cache.get(id(mask_local))
        if hit is None or hit[0] is not mask_local:
            count = self._all_reduce(mask_local.to(torch.float32).sum().reshape(1))
            hit = (mask_local, float(count.item()) / mask_local.numel())
            self._mask_cache[id(mask_local)] = hit
        return hit[1]

    def train_step(self, x_local, labels_local, mask_local):
        """One optimiser step; returns the global loss (data term + weight decay) as a 0-d tensor.
        Plans, mask counts and gathered labels are cached per (labels, mask) tensor OBJECT: pass a new
        tensor for a new label set or mask -- a tensor changed in place keeps its cached plans.
        Loss, accuracy rows and d loss / d logits come from one kernel over the local logits
        (gcn/metrics.py:85-92 with the mask normalised globally) -- over the labelled nodes' logits alone
        when the step runs on their receptive field; the weight gradients and the loss are summed over the
        ranks in one all-reduce; Adam and the weight-decay term run in one kernel per variable."""
        self.step_count += 1
        x_local, labels_local, mask_local = (self._in_model_order(t) for t in (x_local, labels_local, mask_local))
        mask_mean = self._local_mask_mean(mask_local)
        with torch.enable_grad():
            if self.receptive_field and self.aggregate_first and len(self.weights) == 2:
                plans = self._receptive_plans(mask_local)
                logits = self._forward_receptive(x_local, plans)          # one row per labelled local node
                rows = plans['labelled']
                # the gathered labels belong to the pair (labels, mask): the rows come from the mask
                key = ('labels', id(labels_local), id(mask_local))
                hit = self._mask_cache.get(key)
                if hit is None or hit[0] is not labels_local or hit[3] is not mask_local:
                    hit = (labels_local, labels_local[rows].contiguous(),
                           torch.ones(rows.numel(), dtype=torch.int32, device=rows.device), mask_local)
                    self._mask_cache[key] = hit
                if rows.numel():
                    # every labelled row weighs 1 / (global number of labelled nodes): the loss kernel divides
                    # by the rows it is given and by mask_mean
                    count_global = mask_mean * mask_local.numel()
                    stats, dlogits = sparse.softmax_xent(logits.detach(), hit[1], hit[2], count_global / rows.numel())
                    local_loss = stats[:, 0].mean().reshape(1)
                else:                                                     # a rank without labelled nodes
                    dlogits, local_loss = torch.zeros_like(logits), torch.zeros(1, device=logits.device)
            else:
                logits = self._forward(x_local, train=True,
                                       logit_plans=self._logit_plans(mask_local) if self.labelled_logits_only else None)
                stats, dlogits = sparse.softmax_xent(logits.detach(), labels_local, mask_local, mask_mean)
                local_loss = stats[:, 0].mean().reshape(1)
            grads = torch.autograd.grad(logits, self.weights, grad_outputs=dlogits)
        flat = torch.cat([g.reshape(-1) for g in grads] + [local_loss])
        self._all_reduce(flat)
        with torch.no_grad():
            reg = sum(self.wd * (w * w).sum() / 2 for w in self.weights)       # with the weights of this forward
        pos, summed = 0, []
        for w in self.weights:
            summed.append(flat[pos:pos + w.numel()].view_as(w))
            pos += w.numel()
        self.optimizer.t += 1
        self.optimizer.step_fused(summed, self.optimizer.step_size(self.optimizer.t), self.wd)
        return flat[pos] + reg
