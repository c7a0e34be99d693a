"""The fused loss and optimiser kernels against the oracle's restatement of gcn/metrics.py:85-101 and of
TensorFlow's Adam (gcn/models.py:35-38).  PARITY UNPINNED at the TensorFlow boundary (no golden vectors
in the reference); tolerance 1e-5 relative, written at each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(x):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize('n,c', [(2708, 7), (3327, 6), (19717, 3), (1, 1), (100003, 16), (513, 40), (1000, 64), (777, 70)])
def test_masked_cross_entropy_accuracy_and_gradient(oracle, n, c):
    from gcn_amd import sparse
    rng = np.random.default_rng(n + c)
    logits = (rng.standard_normal((n, c)) * 3).astype(np.float32)
    cls = rng.integers(0, c, n)
    labels = np.zeros((n, c), dtype=np.int32)
    labels[np.arange(n), cls] = 1
    labels[rng.random(n) < 0.05] = 0                        # padded nodes: all-zero label rows
    logits[rng.random(n) < 0.3, 0] = 9.0                    # many ties broken towards class 0 never arise; many class-0 wins
    mask = (rng.random(n) < 0.3).astype(np.int32)
    mask[0] = 1
    stats, grad = sparse.softmax_xent(_t(logits), _t(labels), _t(mask), float(mask.astype(np.float32).mean()))
    stats, grad = stats.cpu().numpy(), grad.cpu().numpy()
    ref_loss = float(oracle.GcnOracle.masked_softmax_cross_entropy(logits, labels, mask))
    ref_acc = float(oracle.GcnOracle.masked_accuracy(logits, labels, mask))
    assert abs(stats[:, 0].astype(np.float64).mean() - ref_loss) <= 1e-5 * max(abs(ref_loss), 1e-3)
    assert abs(stats[:, 1].astype(np.float64).mean() - ref_acc) <= 1e-5
    # gradient of the mean loss w.r.t. the logits, in float64
    z = logits.astype(np.float64)
    e = np.exp(z - z.max(1, keepdims=True))
    sm = e / e.sum(1, keepdims=True)
    w = mask / mask.astype(np.float64).mean() / n
    ref_grad = (sm * labels.sum(1, keepdims=True) - labels) * w[:, None]
    assert np.abs(grad - ref_grad).max() <= 1e-5 * np.abs(ref_grad).max()
    assert np.all(grad[mask == 0] == 0)
    # without the gradient
    stats2, none = sparse.softmax_xent(_t(logits), _t(labels), _t(mask), float(mask.astype(np.float32).mean()),
                                       want_grad=False)
    assert none is None and np.array_equal(stats2.cpu().numpy(), stats)


@pytest.mark.parametrize('c', [3, 7, 16, 33, 70])
def test_ties_go_to_the_first_class(c):
    """argmax of equal logits (and of an all-zero label row) is class 0: such rows count as correct."""
    import torch
    from gcn_amd import sparse
    n = 130
    logits = torch.zeros((n, c), device='cuda')
    logits[1::2, c // 2:] = 2.0                              # odd rows: the first maximum sits at c // 2
    labels = torch.zeros((n, c), dtype=torch.int32, device='cuda')
    labels[1::2, c // 2] = 1
    mask = torch.ones(n, dtype=torch.int32, device='cuda')
    stats, _ = sparse.softmax_xent(logits, labels, mask, 1.0)
    assert bool((stats[:, 1] == 1.0).all())
    labels[1::2] = 0
    labels[1::2, c - 1] = 1                                  # now the label is the LAST of the tied maxima
    stats, _ = sparse.softmax_xent(logits, labels, mask, 1.0)
    assert bool((stats[1::2, 1] == 0.0).all()) and bool((stats[0::2, 1] == 1.0).all())


def test_loss_kernel_rejects_an_empty_mask():
    import torch
    from gcn_amd import sparse
    with pytest.raises(ValueError):
        sparse.softmax_xent(torch.zeros((4, 3), device='cuda'), torch.zeros((4, 3), dtype=torch.int32, device='cuda'),
                            torch.zeros(4, dtype=torch.int32, device='cuda'), 0.0)


@pytest.mark.parametrize('shape', [(1433, 16), (16, 7), (1,), (100001,)])
def test_adam_steps_follow_the_oracle(oracle, shape):
    """Five steps with weight decay folded in, step size passed by value and read from device memory."""
    import torch
    from gcn_amd import sparse
    rng = np.random.default_rng(7)
    w0 = rng.standard_normal(shape).astype(np.float32)
    wd = np.float32(5e-4)
    for on_device in (False, True):
        ref_w = [w0.copy()]
        adam = oracle.TfAdam(ref_w, 0.02)
        p, m, v = _t(w0.copy()), torch.zeros(shape, device='cuda'), torch.zeros(shape, device='cuda')
        lr_dev = torch.zeros((), device='cuda')
        for t in range(1, 6):
            g = rng.standard_normal(shape).astype(np.float32) * 0.1
            adam.step(ref_w, [g + wd * ref_w[0]])
            lr_t = 0.02 * float(np.sqrt(1.0 - 0.999 ** t)) / (1.0 - 0.9 ** t)
            if on_device:
                lr_dev.fill_(lr_t)
            sparse.adam_step(p, _t(g), m, v, lr_dev if on_device else lr_t, weight_decay=float(wd))
        assert np.abs(p.cpu().numpy() - ref_w[0]).max() <= 1e-5 * np.abs(ref_w[0]).max()
        assert np.abs(m.cpu().numpy() - adam.m[0]).max() <= 1e-6 * max(np.abs(adam.m[0]).max(), 1e-30)


@pytest.mark.parametrize('n,n_w', [(2708, 23040), (1, 0), (100003, 7), (3_000_000, 70000)])
def test_loss_summary_is_the_means_plus_weight_decay(n, n_w):
    """out = [mean(stats[:, 0]) + wd * sum(w^2) / 2, mean(stats[:, 1])] (gcn/models.py:92-99, 106-108), one
    block or several with the partial sums added in a fixed order; tolerance 1e-6 relative (float64 sums)."""
    import torch
    from gcn_amd import sparse
    gen = torch.Generator().manual_seed(n)
    stats = torch.rand((n, 2), generator=gen).cuda()
    w = (torch.rand(n_w, generator=gen) - 0.5).cuda() if n_w else None
    wd = 5e-4
    out = sparse.loss_summary(stats, w, wd)
    ref_loss = float(stats[:, 0].double().mean()) + (wd * float((w.double() ** 2).sum()) / 2 if n_w else 0.0)
    ref_acc = float(stats[:, 1].double().mean())
    got_loss, got_acc = out.tolist()
    assert abs(got_loss - ref_loss) <= 1e-6 * abs(ref_loss) and abs(got_acc - ref_acc) <= 1e-6 * abs(ref_acc)
    assert torch.equal(out, sparse.loss_summary(stats, w, wd))          # deterministic
