"""GPU parity of the layer / model / driver mirror (gcn/layers.py, gcn/models.py, train.py) against
the CPU oracle on Cora and CiteSeer: logits, loss, accuracy, gradients, Adam steps, dropout masks.

PARITY UNPINNED at the TensorFlow boundary (TensorFlow is not importable and the reference holds
no golden vectors): the checker is the oracle's restatement of the cited lines.  Tolerance for
float32 tensors: max|x - ref| <= 1e-5 * max|ref| (the sparse products themselves are bit-exact;
the dense H.W product runs on the MFMA pipe in a different association order than numpy).
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _close(a, b, tol=REL_TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30)


def _setup(name, oracle, seed=5):
    from gcn_amd import graph as tf, sparse
    from gcn_amd.models import GCN_MLP
    from gcn_amd.utils import SupportValue, construct_feed_dict
    import scipy.sparse as sp
    from gcn_amd.utils import preprocess_adj
    fx = load_golden(name)
    n = int(fx['n'])
    adj = sp.csr_matrix((np.ones(fx['adj_row'].size), (fx['adj_row'], fx['adj_col'])), shape=(n, n))
    support = [preprocess_adj(adj)]
    features = SupportValue(fx['feat_coords'], fx['feat_values'], fx['feat_shape'])
    y_train, train_mask = fx['y_train'].astype(np.float64), fx['train_mask']
    tf.set_random_seed(seed)
    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32, name='support0')],
        'features': tf.sparse_placeholder(tf.float32, name='features'),
        'labels': tf.placeholder(tf.int32, name='labels', shape=(None, y_train.shape[1])),
        'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
        'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
    }
    cfg = {'name': 'gcn', 'connection': ['c', 'c'], 'layer_size': [16], 'conv': 'gcn', 'learning_rate': 0.02,
           'weight_decay': 5e-4, 'Model': 'GCN', 'dropout': 0.5}
    model = GCN_MLP(cfg, placeholders, input_dim=int(fx['feat_shape'][1]))
    feed = construct_feed_dict(features, support, y_train, train_mask, placeholders)
    ip, ix, v32, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    feat = (fx['feat_indptr'].astype(np.int64), fx['feat_coords'][:, 1], fx['feat_values'].astype(np.float32))
    weights = [w.detach().cpu().numpy().copy() for w in model._trainable()]
    ref = oracle.GcnOracle((ip, ix, v32), feat, n, weights)
    return fx, tf, model, placeholders, feed, ref, support


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_preprocess_adj_triple_and_device_csr(name, oracle):
    """The drop-in preprocess_adj: triple as the reference returns it + the resident CSR."""
    fx, tf, model, ph, feed, ref, support = _setup(name, oracle)
    s = support[0]
    indices, values, dense_shape = s                                   # callers unpack it (train.py:125)
    assert indices.dtype == np.int32 and np.array_equal(indices, fx['hat_indices'])
    assert np.array_equal(dense_shape, fx['hat_shape'])
    assert np.abs(values.view(np.int64) - fx['hat_values'].view(np.int64)).max() <= 4      # float64 ulps
    assert np.array_equal(s.csr.indptr.cpu().numpy(), fx['ai_indptr'])
    assert np.array_equal(s.csr.indices.cpu().numpy(), fx['ai_indices'])
    assert np.array_equal(s.csr.values.cpu().numpy(), fx['hat_values'].astype(np.float32))


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_forward_loss_accuracy_at_fixed_weights(name, oracle):
    fx, tf, model, ph, feed, ref, _ = _setup(name, oracle)
    sess = tf.Session()
    outputs, loss, acc, acc_cls, pred = sess.run(
        [model.outputs, model.loss, model.accuracy, model.accuracy_of_class, model.prediction], feed_dict=feed)
    logits = ref.forward()
    labels, mask = fx['y_train'], fx['train_mask']
    assert _close(outputs, logits)
    # layer 1 is sparse-only arithmetic: bit-exact
    assert np.array_equal(model.activations[1].cpu().numpy(), np.maximum(ref.cache['z'][0], 0))
    assert abs(float(loss) - float(ref.loss(logits, labels, mask))) <= 1e-5 * abs(float(loss))
    assert abs(float(acc) - float(ref.masked_accuracy(logits, labels, mask))) <= 1e-6
    for i, a in enumerate(acc_cls):
        m = mask * labels[:, i]
        assert abs(float(a) - float(ref.masked_accuracy(logits, labels, m))) <= 1e-6
    e = np.exp(logits - logits.max(1, keepdims=True))
    assert _close(pred, e / e.sum(1, keepdims=True))
    assert outputs.shape == (int(fx['n']), labels.shape[1])


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_gradients_and_adam_steps(name, oracle):
    import torch
    fx, tf, model, ph, feed, ref, _ = _setup(name, oracle)
    labels, mask = fx['y_train'], fx['train_mask']
    # raw gradients, dropout off
    model._bind(feed)
    from gcn_amd import sparse
    with torch.enable_grad():
        out = model._forward()
        dev_mask = ph['labels_mask'].value
        stats, dlogits = sparse.softmax_xent(out.detach(), ph['labels'].value, dev_mask, model._mask_mean(dev_mask))
        grads = torch.autograd.grad(out, model._trainable(), grad_outputs=dlogits)
    logits = ref.forward()
    ref_grads = ref.backward(logits, labels, mask)
    for g, w, rg in zip(grads, model._trainable(), ref_grads):     # the optimiser kernel adds weight_decay * w
        assert _close((g + 5e-4 * w.detach()).cpu().numpy(), rg, 2e-5)
    assert abs(float(stats[:, 0].mean()) - float(ref.masked_softmax_cross_entropy(logits, labels, mask))) <= 1e-6
    assert abs(float(stats[:, 1].mean()) - float(ref.masked_accuracy(logits, labels, mask))) <= 1e-6
    # three optimiser steps (TensorFlow's Adam form), dropout off
    sess = tf.Session()
    adam = oracle.TfAdam(ref.w, 0.02)
    for _ in range(3):
        sess.run(model.opt_op, feed_dict=feed)
        lg = ref.forward()
        adam.step(ref.w, ref.backward(lg, labels, mask))
    # Adam divides by sqrt(v): where a gradient component passes through zero it turns float32
    # rounding noise into an O(lr * 1e-3) difference, so ask 1e-5 of all but a few entries and
    # 1e-3 of every entry.
    for p, rw in zip(model._trainable(), ref.w):
        d = np.abs(p.detach().cpu().numpy().astype(np.float64) - rw)
        scale = np.abs(rw).max()
        assert (d <= REL_TOL * scale).mean() >= 0.999
        assert d.max() <= 1e-3 * scale


def test_training_mode_uses_the_same_dropout_masks_as_the_oracle(oracle):
    fx, tf, model, ph, feed, ref, _ = _setup('cora', oracle, seed=77)
    feed = dict(feed)
    feed[ph['dropout']] = 0.5
    sess = tf.Session()
    outputs, loss = sess.run([model.outputs, model.loss], feed_dict=feed)
    run = model._run_counter
    logits = ref.forward(dropout=0.5, seed=77, stream_ids=[run * 64 + 0, run * 64 + 1])
    # the first layer's product stores its ReLU output with the second layer's input dropout applied:
    # the oracle's dropped hidden matrix, bit for bit
    assert np.array_equal(model.activations[1].cpu().numpy(), ref.cache['hin'][1])
    assert _close(outputs, logits)
    # about half of X's entries were dropped, and a second run draws a different mask
    assert 0.45 < ref.cache['xmask'].mean() < 0.55
    outputs2 = sess.run(model.outputs, feed_dict=feed)
    assert not np.array_equal(outputs, outputs2)
    # evaluation feeds leave dropout at its default 0 (train.py:149, 182-183)
    feed_eval = {k: v for k, v in feed.items() if k is not ph['dropout']}
    a = sess.run(model.outputs, feed_dict=feed_eval)
    b = sess.run(model.outputs, feed_dict=feed_eval)
    assert np.array_equal(a, b)


def test_short_training_trajectory_follows_the_oracle(oracle):
    """Five epochs with dropout 0.5: identical masks (counter-based RNG) and the same Adam, so the
    loss curve of the GPU model tracks the CPU oracle's."""
    fx, tf, model, ph, feed, ref, _ = _setup('cora', oracle, seed=3)
    labels, mask = fx['y_train'], fx['train_mask']
    feed = dict(feed)
    feed[ph['dropout']] = 0.5
    sess = tf.Session()
    adam = oracle.TfAdam(ref.w, 0.02)
    for epoch in range(5):
        sess.run(model.opt_op, feed_dict=feed)
        run = model._run_counter
        lg = ref.forward(dropout=0.5, seed=3, stream_ids=[run * 64, run * 64 + 1])
        adam.step(ref.w, ref.backward(lg, labels, mask))
        loss = sess.run(model.loss, feed_dict=feed)
        run = model._run_counter
        ref_loss = ref.loss(ref.forward(dropout=0.5, seed=3, stream_ids=[run * 64, run * 64 + 1]), labels, mask)
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss)), (epoch, loss, ref_loss)


@pytest.mark.parametrize('name,floor', [('cora', 0.72), ('citeseer', 0.60)])
def test_driver_reaches_the_expected_accuracy(name, floor, capsys):
    """train.train() end to end from the raw dataset files: 20 labels per class, 200 epochs, test on
    the best-train-accuracy epoch.  Kipf-style GCNs reach ~0.80 / ~0.68 on these random splits."""
    import train as driver
    from config import load_configuration
    from gcn_amd import graph as tf
    configuration, args = load_configuration(['--dataset', name, '--train_size', '20'], echo=False)
    driver.args = args
    cfg = configuration['model_list'][0]
    seed = 123
    np.random.seed(seed)
    tf.set_random_seed(seed)
    with tf.Session() as sess:
        test_acc, test_acc_of_class, prediction, seconds = driver.train(cfg, sess, seed)
    assert floor <= float(test_acc) <= 1.0
    assert prediction.shape[1] == len(test_acc_of_class)
    assert np.allclose(prediction.sum(1), 1.0, atol=1e-5)
    assert 'Optimization Finished!' in capsys.readouterr().out


def test_recorded_train_step_follows_the_eager_one():
    """The train step replayed from a HIP graph (dropout stream offset and Adam step size read from
    device memory) against the step launched kernel by kernel, on Cora with dropout 0.5: the masks are
    the same, so loss and accuracy agree step for step and the weights to rounding of the Adam update."""
    from config import load_configuration
    from gcn_amd import graph as tf, models, utils

    def trajectory(extra, steps=6):
        configuration, _ = load_configuration(['--dataset', 'cora', '--train_size', '20'] + extra, echo=False)
        cfg = configuration['model_list'][0]
        np.random.seed(123)
        tf.set_random_seed(123)
        adj, features, y_train, _, _, train_mask, _, _ = utils.load_data(
            'cora', train_size=20, validation_size=500, model_config=cfg, shuffle=True)
        support = [utils.preprocess_adj(adj)]
        if not isinstance(features, tuple):
            features = utils.sparse_to_tuple(utils.preprocess_features(features, 'bow'))
        placeholders = {
            'support': [tf.sparse_placeholder(tf.float32, name='support_0')],
            'features': tf.sparse_placeholder(tf.float32, name='features'),
            'labels': tf.placeholder(tf.int32, shape=(None, y_train.shape[1]), name='labels'),
            'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
            'dropout': tf.placeholder_with_default(0., shape=(), name='dropout'),
            'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero')}
        model = models.GCN_MLP(cfg, placeholders, input_dim=features[2][1])
        feed = utils.construct_feed_dict(features, support, y_train, train_mask, placeholders)
        feed.update({placeholders['dropout']: 0.5})
        sess = tf.Session()
        seen = []
        for _ in range(steps):
            _, loss, acc = sess.run([model.opt_op, model.loss, model.accuracy], feed_dict=feed)
            seen.append((float(loss), float(acc)))
            sess.run([model.loss, model.accuracy], feed_dict=feed)      # an eager run between replays, as train.py makes
        return np.array(seen), [v.detach().cpu().numpy() for v in model._trainable()], model

    eager, w_eager, _ = trajectory([])
    recorded, w_recorded, model = trajectory(['--hip_graph'])
    assert len(model._recorded) == 1                                     # one capture, replayed
    assert np.allclose(eager, recorded, rtol=1e-5, atol=1e-6)
    assert eager[-1, 0] < eager[0, 0]
    for a, b in zip(w_eager, w_recorded):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_test_accuracy_over_seeds_matches_the_oracle(name, oracle):
    """10 seeds (weight init + dropout streams), 60 epochs each, the GPU model and the CPU oracle trained
    side by side on the fixture's split: the test accuracies have the same mean to within one standard
    deviation (they are in fact nearly equal seed by seed: same masks, same Adam)."""
    gpu_acc, ref_acc = [], []
    for seed in range(1, 11):
        fx, tf, model, ph, feed, ref, _ = _setup(name, oracle, seed=seed)
        labels, mask = fx['y_train'], fx['train_mask']
        feed = dict(feed)
        feed[ph['dropout']] = 0.5
        test_feed = dict(feed)
        test_feed[ph['labels']] = fx['y_test'].astype(np.float64)
        test_feed[ph['labels_mask']] = fx['test_mask']
        test_feed[ph['dropout']] = 0.0
        sess = tf.Session()
        adam = oracle.TfAdam(ref.w, 0.02)
        for epoch in range(60):
            sess.run(model.opt_op, feed_dict=feed)
            run = model._run_counter
            lg = ref.forward(dropout=0.5, seed=seed, stream_ids=[run * 64, run * 64 + 1])
            adam.step(ref.w, ref.backward(lg, labels, mask))
        gpu_acc.append(float(sess.run(model.accuracy, feed_dict=test_feed)))
        ref_acc.append(float(ref.masked_accuracy(ref.forward(), fx['y_test'], fx['test_mask'])))
    gpu_acc, ref_acc = np.array(gpu_acc), np.array(ref_acc)
    sigma = max(ref_acc.std(), 1e-3)
    assert abs(gpu_acc.mean() - ref_acc.mean()) <= sigma, (gpu_acc, ref_acc)
    assert np.abs(gpu_acc - ref_acc).max() <= 0.02, (gpu_acc, ref_acc)
    assert ref_acc.mean() > (0.70 if name == 'cora' else 0.58)


@pytest.mark.parametrize('dense_input', [False, True])
def test_pubmed_two_layer_logits_at_fixed_weights(oracle, dense_input):
    """PubMed (19,717 nodes, 3 classes): A_hat from the reference's own preprocess_adj (fixture), X a
    seeded synthetic 19,717 x 500 matrix -- the dataset's feature file is absent upstream -- fed either
    as a sparse triple (entries above 0.8 kept: ~50 per row, like the real bag of words) or as a dense
    placeholder; fixed Glorot weights, dropout 0 (gcn/models.py:54-69).
    Criterion: max|gpu - ref64| <= 1e-5 * max|ref64| against a float64 evaluation of the same model."""
    import scipy.sparse as sp
    from gcn_amd import graph as tf
    from gcn_amd.models import GCN_MLP
    from gcn_amd.utils import SupportValue, construct_feed_dict
    fx = load_golden('pubmed_graph')
    n, f_in, classes = int(fx['n']), 500, 3
    x = oracle.synthetic_dense(n, f_in, seed=19)
    if not dense_input:
        x = np.where(x > 0.8, x, np.float32(0))
    support = [SupportValue(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])]
    rng = np.random.RandomState(8)
    labels = np.eye(classes)[rng.randint(0, classes, n)]
    mask = np.zeros(n, dtype=bool)
    mask[rng.choice(n, 60, replace=False)] = True
    tf.set_random_seed(41)
    if dense_input:
        features = x
        feat_ph = tf.placeholder(tf.float32, shape=(None, f_in), name='features')
    else:
        r, c = np.nonzero(x)                                   # row-major, columns ascending
        features = SupportValue(np.stack([r, c], 1).astype(np.int32), x[r, c], np.array([n, f_in], dtype=np.int64))
        feat_ph = tf.sparse_placeholder(tf.float32, name='features')
        assert 40 < r.size / n < 60
    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32, name='support0')],
        'features': feat_ph,
        'labels': tf.placeholder(tf.int32, name='labels', shape=(None, classes)),
        'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
        'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
    }
    cfg = {'name': 'gcn', 'connection': ['c', 'c'], 'layer_size': [16], 'conv': 'gcn', 'learning_rate': 0.02,
           'weight_decay': 5e-4, 'Model': 'GCN', 'dropout': 0.5}
    model = GCN_MLP(cfg, placeholders, input_dim=f_in)
    if dense_input:
        feed = {placeholders['labels']: labels, placeholders['labels_mask']: mask, placeholders['features']: x,
                placeholders['support'][0]: support[0], placeholders['num_features_nonzero']: (0,)}
    else:
        feed = construct_feed_dict(features, support, labels, mask, placeholders)
    outputs, loss, acc = tf.Session().run([model.outputs, model.loss, model.accuracy], feed_dict=feed)
    # float64 evaluation with the float32 values the placeholders hold (train.py:145-146)
    w0, w1 = (w.detach().cpu().numpy().astype(np.float64) for w in model._trainable())
    a = sp.csr_matrix((fx['hat_values'].astype(np.float32).astype(np.float64),
                       (fx['hat_indices'][:, 0], fx['hat_indices'][:, 1])), shape=(n, n))
    h = np.maximum(a @ (x.astype(np.float64) @ w0), 0)
    ref64 = a @ (h @ w1)
    assert outputs.shape == (n, classes)
    assert np.abs(outputs - ref64).max() <= REL_TOL * np.abs(ref64).max()
    ref_loss = 5e-4 * ((w0 ** 2).sum() + (w1 ** 2).sum()) / 2 \
        + float(oracle.GcnOracle.masked_softmax_cross_entropy(ref64, labels, mask))      # gcn/models.py:92-99
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    assert abs(float(acc) - float(oracle.GcnOracle.masked_accuracy(ref64, labels, mask))) <= 1e-6


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
@pytest.mark.parametrize('reorder', ['degree', 'hubs', ('hubs', 100)])
def test_model_on_a_relabelled_support_equals_the_oracle_on_the_relabelled_graph(name, reorder, oracle):
    """preprocess_adj(reorder=...): the model multiplies with the matrix of the relabelled graph and permutes
    features, labels and mask in and its outputs back.  The oracle, run on the relabelled matrix with the
    permuted inputs, must agree: first-layer activations (sparse products only) bit for bit, logits, loss
    and three training steps with dropout (decisions drawn in the relabelled order on both sides) to the
    float32 tolerance -- and the triple the caller sees stays the reference's."""
    import scipy.sparse as sp
    import torch
    from gcn_amd import graph as tf
    from gcn_amd.models import GCN_MLP
    from gcn_amd.utils import SupportValue, construct_feed_dict, preprocess_adj
    fx = load_golden(name)
    n = int(fx['n'])
    adj = sp.csr_matrix((np.ones(fx['adj_row'].size), (fx['adj_row'], fx['adj_col'])), shape=(n, n))
    support = [preprocess_adj(adj, reorder=reorder)]
    s = support[0]
    assert np.array_equal(s.indices, fx['hat_indices']) and np.array_equal(s.dense_shape, fx['hat_shape'])
    new_id, old_id = s.new_id.cpu().numpy(), s.old_id.cpu().numpy()
    ip0, ix0, v0, _ = oracle.triple_to_csr(fx['hat_indices'], fx['hat_values'], fx['hat_shape'])
    want = oracle.degree_order(ip0) if reorder == 'degree' else oracle.hub_order(ip0, 65536 if reorder == 'hubs' else 100)
    assert np.array_equal(new_id, want[0]) and np.array_equal(old_id, want[1])
    rip, rix, rv = oracle.csr_relabel(n, ip0, ix0, v0.astype(np.float64), new_id, new_id)
    assert np.array_equal(s.csr.indptr.cpu().numpy(), rip) and np.array_equal(s.csr.indices.cpu().numpy(), rix)
    assert np.array_equal(s.csr.values.cpu().numpy(), rv.astype(np.float32))
    # the model, fed in the dataset's order
    features = SupportValue(fx['feat_coords'], fx['feat_values'], fx['feat_shape'])
    y_train, train_mask = fx['y_train'].astype(np.float64), fx['train_mask']
    tf.set_random_seed(9)
    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32, name='support0')],
        'features': tf.sparse_placeholder(tf.float32, name='features'),
        'labels': tf.placeholder(tf.int32, name='labels', shape=(None, y_train.shape[1])),
        'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
        'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
    }
    cfg = {'name': 'gcn', 'connection': ['c', 'c'], 'layer_size': [16], 'conv': 'gcn', 'learning_rate': 0.02,
           'weight_decay': 5e-4, 'Model': 'GCN', 'dropout': 0.5}
    model = GCN_MLP(cfg, placeholders, input_dim=int(fx['feat_shape'][1]))
    feed = construct_feed_dict(features, support, y_train, train_mask, placeholders)
    # the oracle on the relabelled problem: X, labels and mask with their rows in the new order
    fip = fx['feat_indptr'].astype(np.int64)
    lens = np.diff(fip)[old_id]
    pip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    src = np.repeat(fip[:-1][old_id] - pip[:-1], lens) + np.arange(pip[-1])
    feat = (pip, fx['feat_coords'][:, 1][src], fx['feat_values'].astype(np.float32)[src])
    weights = [w.detach().cpu().numpy().copy() for w in model._trainable()]
    ref = oracle.GcnOracle((rip, rix, rv.astype(np.float32)), feat, n, weights)
    labels_r, mask_r = fx['y_train'][old_id], fx['train_mask'][old_id]
    sess = tf.Session()
    outputs, loss = sess.run([model.outputs, model.loss], feed_dict=feed)
    logits = ref.forward()
    assert np.array_equal(model.activations[1].cpu().numpy(), np.maximum(ref.cache['z'][0], 0))
    assert _close(outputs, logits[new_id])                                # outputs come back in the dataset's order
    assert abs(float(loss) - float(ref.loss(logits, labels_r, mask_r))) <= 1e-5 * abs(float(loss))
    # and they are the unordered model's outputs up to the order of the additions
    plain = [preprocess_adj(adj)]
    feed_plain = construct_feed_dict(features, plain, y_train, train_mask, placeholders)
    assert _close(sess.run(model.outputs, feed_dict=feed_plain), outputs, 1e-5)
    # three optimiser steps with dropout
    feed = dict(feed)
    feed[placeholders['dropout']] = 0.5
    adam = oracle.TfAdam(ref.w, 0.02)
    for _ in range(3):
        sess.run(model.opt_op, feed_dict=feed)
        run = model._run_counter
        lg = ref.forward(dropout=0.5, seed=9, stream_ids=[run * 64, run * 64 + 1])
        adam.step(ref.w, ref.backward(lg, labels_r, mask_r))
    for p, rw in zip(model._trainable(), ref.w):
        d = np.abs(p.detach().cpu().numpy().astype(np.float64) - rw)
        scale = np.abs(rw).max()
        assert (d <= REL_TOL * scale).mean() >= 0.999 and d.max() <= 1e-3 * scale
