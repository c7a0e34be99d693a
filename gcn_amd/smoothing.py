"""Repeated propagation on a resident support: A_hat^k X and the absorption-probability
approximation of the reference (gcn/utils.py:559-571), each step one fused HIP product."""
import numpy as np
import scipy.sparse as sp
import torch

from . import sparse
from .utils import _normalize_on_device


def propagate(a, x, k):
    """A^k X for a DeviceCSR ``a`` and dense float32 ``x``, ping-ponging two buffers."""
    cur = x if x.dtype == torch.float32 else x.float()
    bufs = [torch.empty_like(cur), torch.empty_like(cur)]
    for step in range(k):
        nxt = bufs[step & 1]
        sparse.spmm(a, cur, out=nxt)
        cur = nxt
    return cur if k > 0 else cur.clone()


class GraphedPropagation(object):
    """``k`` propagation steps recorded once as a HIP graph and replayed with a single launch.

    On the citation graphs one product takes a few microseconds of GPU time, so ``k`` separate
    launches are bound by the host's launch rate; a graph replays all of them back to back.
    ``addend=True`` records ``new = A . new + X`` starting from zero (the update of
    gcn/utils.py:566-569) instead of ``A^k X``.  Buffers are owned by the object: ``__call__``
    copies ``x`` into the recorded input and returns the recorded output (overwritten by the next
    call)."""

    def __init__(self, a, n_features, k, addend=False):
        if k < 1:
            raise ValueError('k must be at least 1')
        self.a, self.k, self.addend = a, int(k), bool(addend)
        dev = a.device
        self.x = torch.zeros((a.n_cols, n_features), dtype=torch.float32, device=dev)
        self._zero = torch.zeros_like(self.x) if addend else None
        self._bufs = [torch.empty_like(self.x), torch.empty_like(self.x)]
        self._ws = a.plan.workspace(n_features, dev)          # allocated before the capture, kept alive with it
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                          # one eager pass: everything lazy happens here
            self._steps()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._steps()

    def _steps(self):
        cur = self._zero if self.addend else self.x
        for step in range(self.k):
            nxt = self._bufs[step & 1]
            if self.addend:
                sparse.spmm_add(self.a, cur, self.x, out=nxt)
            else:
                sparse.spmm(self.a, cur, out=nxt)
            cur = nxt
        return cur

    def __call__(self, x):
        self.x.copy_(x)
        self.graph.replay()
        return self.out


def ap_approximate(adj, features, alpha, k, device='cuda'):
    """gcn/utils.py:559-571: adj' = rownorm(adj + I) / (alpha + 1); new = 0; k times
    new = adj' . new + features; result = new * alpha / (alpha + 1).  The reference iterates in
    float64 on the host; this runs in float32 on the GPU with the `+ features` fused into the product."""
    n, indptr, indices, w64, _ = _normalize_on_device(adj, 'rw', True, device)
    vals = (w64 / (alpha + 1.0)).to(torch.float32)
    a = sparse.DeviceCSR(n, n, indptr, indices, vals)
    if sp.issparse(features):
        features = features.toarray()
    x = torch.as_tensor(np.asarray(features, dtype=np.float32)).to(device)
    cur = torch.zeros_like(x)
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    for step in range(k):
        nxt = bufs[step & 1]
        sparse.spmm_add(a, cur, x, out=nxt)
        cur = nxt
    return (cur * (alpha / (alpha + 1.0))).cpu().numpy()
