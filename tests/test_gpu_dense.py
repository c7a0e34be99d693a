"""GPU parity of the dense pieces of the layer: tf.nn.dropout (gcn/layers.py:185) and
tf.matmul (gcn/layers.py:32-33) on the float32 MFMA pipe."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _t(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('keep_prob', [0.5, 0.8, 1.0])
def test_dense_dropout_bit_exact(oracle, keep_prob):
    from gcn_amd import sparse
    x = oracle.synthetic_dense(2708, 16, seed=31)
    ref, keep = oracle.dense_dropout(x, keep_prob, oracle.stream_key(99, 3))
    out = sparse.dropout(_t(x), keep_prob, 99, 3).cpu().numpy()
    assert np.array_equal(out, ref)
    # with a gate: the ReLU-backward form
    gate = oracle.synthetic_dense(2708, 16, seed=32)
    out_g = sparse.dropout(_t(x), keep_prob, 99, 3, gate=_t(gate)).cpu().numpy()
    assert np.array_equal(out_g, np.where(gate > 0, ref, np.float32(0)))


@pytest.mark.parametrize('m,n,k', [(2708, 7, 16), (3327, 6, 16), (19717, 3, 16), (1000, 16, 1433), (513, 130, 257),
                                   (4096, 512, 512), (1, 1, 1), (129, 33, 2)])
@pytest.mark.parametrize('trans_b', [False, True])
def test_gemm_mfma_against_float64(oracle, m, n, k, trans_b):
    from gcn_amd import sparse
    a = oracle.synthetic_dense(m, k, seed=41)
    b = oracle.synthetic_dense(k, n, seed=42)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bt = _t(np.ascontiguousarray(b.T)) if trans_b else _t(b)
    out = sparse.gemm(_t(a), bt, trans_b=trans_b).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= REL_TOL * scale
    out_relu = sparse.gemm(_t(a), bt, trans_b=trans_b, relu=True).cpu().numpy()
    assert np.array_equal(out_relu, np.maximum(out, 0))


def test_gemm_identity_with_asymmetric_operand():
    """A = I against an asymmetric B catches a swapped row/column map of the MFMA result."""
    import torch
    from gcn_amd import sparse
    n = 96
    b = torch.arange(n * 40, dtype=torch.float32, device='cuda').reshape(n, 40)
    out = sparse.gemm(torch.eye(n, device='cuda'), b)
    assert torch.equal(out, b)
