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
