"""Graph-side helpers of the reference (gcn/utils.py) with the heavy lifting on the GPU.

``preprocess_adj`` keeps returning the ``(indices, values, dense_shape)`` triple the
reference's callers unpack (train.py:125, utils.py:596); the same object also carries the
device-resident CSR (``.csr``) that the layers multiply with, so the matrix is uploaded and
analysed once instead of being re-fed every step.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

from . import sparse
from .graph import SparseTensorValue


class SupportValue(SparseTensorValue):
    """SparseTensorValue that can also carry the device CSR of the same matrix (``.csr``)."""


def sparse_to_tuple(sparse_mx):
    """Convert a scipy sparse matrix (or a list of them) to the triple representation
    (gcn/utils.py:258-275): ``indices`` [nnz, 2], ``values`` [nnz], ``dense_shape`` int64 [2]."""

    def to_tuple(mx):
        if not sp.isspmatrix_coo(mx):
            mx = mx.tocoo()
        coords = np.vstack((mx.row, mx.col)).transpose()
        return SupportValue(coords, mx.data, np.array(mx.shape, dtype=np.int64))

    if isinstance(sparse_mx, list):
        for i in range(len(sparse_mx)):
            sparse_mx[i] = to_tuple(sparse_mx[i])
    else:
        sparse_mx = to_tuple(sparse_mx)
    return sparse_mx


def _device_coo(adj, device):
    coo = sp.coo_matrix(adj)
    rows = torch.as_tensor(coo.row.astype(np.int32)).to(device)
    cols = torch.as_tensor(coo.col.astype(np.int32)).to(device)
    vals = torch.as_tensor(coo.data.astype(np.float64)).to(device)
    return coo.shape, rows, cols, vals


def _normalize_on_device(adj, type, loop, device):
    """The kernels behind normalize_adj / preprocess_adj.  Returns (n, indptr, indices, w64, w32) of
    the CSR of (adj [+ I]) with w = the normalised value of every stored entry (see gcn_hip.h)."""
    (n_rows, n_cols), rows, cols, vals = _device_coo(adj, device)
    if n_rows != n_cols:
        raise ValueError('adjacency matrix must be square')
    indptr, indices, data = sparse.coo_to_csr(n_rows, n_cols, rows, cols, vals, add_self_loops=loop)
    d = sparse.degree_scale(sparse.row_sums(indptr, data), type)
    w64, w32 = sparse.normalize_values(indptr, indices, data, d, type)
    return n_rows, indptr, indices, w64, w32


def _rows_of(indptr):
    counts = indptr[1:] - indptr[:-1]
    return torch.repeat_interleave(torch.arange(counts.numel(), device=indptr.device), counts)


def normalize_adj(adj, type='sym', device='cuda'):
    """Symmetrically (or random-walk) normalise an adjacency matrix (gcn/utils.py:297-315);
    returns a scipy matrix like the reference does."""
    if type not in ('sym', 'rw'):
        raise ValueError('Invalid normalisation type: ' + str(type))
    n, indptr, indices, w64, _ = _normalize_on_device(adj, type, False, device)
    rows = _rows_of(indptr).cpu().numpy()
    cols = indices.cpu().numpy()
    if type == 'sym':       # stored entry (r, c) carries result[c, r]
        return sp.coo_matrix((w64.cpu().numpy(), (cols, rows)), shape=(n, n))
    return sp.csr_matrix((w64.cpu().numpy(), cols, indptr.cpu().numpy()), shape=(n, n))


def preprocess_adj(adj, type='sym', loop=True, device='cuda'):
    """Preprocessing of the adjacency matrix for the GCN model and conversion to the triple
    representation (gcn/utils.py:318-323): ``adj + I`` (ADDS 1 to the diagonal), degree^-1/2
    scaling in float64, triple in the element order of the reference (column-major for 'sym').
    The returned triple's ``.csr`` is the float32 device CSR of the same matrix."""
    if type not in ('sym', 'rw'):
        raise ValueError('Invalid normalisation type: ' + str(type))
    n, indptr, indices, w64, w32 = _normalize_on_device(adj, type, loop, device)
    rows = _rows_of(indptr)
    if type == 'sym':
        coords = torch.stack([indices, rows.to(torch.int32)], dim=1)
        # the stored arrays describe result^T; one stable transpose gives the CSR of the result
        # (for an undirected graph it returns the same arrays)
        csr = sparse.DeviceCSR(n, n, indptr, indices, w32).transpose()
        csr.perm = None
        csr._t = sparse.DeviceCSR(n, n, indptr, indices, w32)
    else:
        coords = torch.stack([rows.to(torch.int32), indices], dim=1)
        csr = sparse.DeviceCSR(n, n, indptr, indices, w32)
    out = SupportValue(coords.cpu().numpy(), w64.cpu().numpy(), np.array([n, n], dtype=np.int64))
    out.csr = csr
    return out


def construct_feed_dict(features, support, labels, labels_mask, placeholders):
    """Construct the feed dictionary (gcn/utils.py:590-598)."""
    feed_dict = dict()
    feed_dict.update({placeholders['labels']: labels})
    feed_dict.update({placeholders['labels_mask']: labels_mask})
    feed_dict.update({placeholders['features']: features})
    feed_dict.update({placeholders['support'][i]: support[i] for i in range(len(support))})
    feed_dict.update({placeholders['num_features_nonzero']: features[1].shape})
    return feed_dict


def preprocess_model_config(model_config):
    """The checks of gcn/utils.py:601-616 that gate the GCN path, and the generated model name
    (utils.py:619-640, 689) for the convolution types this build knows."""
    model_config['connection'] = list(model_config['connection'])
    for c in model_config['connection']:
        if c not in ['c', 'd', 'r', 'f', 'C']:
            raise ValueError(
                'connection string specified by --connection can only contain "c", "d", "r", "f", "C" '
                'but "{}" found'.format(c))
    for i in model_config['layer_size']:
        if not isinstance(i, int):
            raise ValueError('layer_size should be a list of int, but found {}'.format(model_config['layer_size']))
        if i <= 0:
            raise ValueError('layer_size must be greater than 0, but found {}'.format(i))
    if not len(model_config['connection']) == len(model_config['layer_size']) + 1:
        raise ValueError('length of connection string should be equal to length of layer_size list plus 1')
    if not model_config.get('name'):
        name = str(model_config['Model']) + '_' + model_config['connection'][0]
        for char, size in zip(model_config['connection'][1:], model_config['layer_size']):
            name += str(size) + char
        if model_config['conv'] in ('gcn_unnorm', 'gcn_noloop'):
            name += '_' + model_config['conv']
        if model_config.get('validate'):
            name += '_validate'
        model_config['name'] = name


# --------------------------------------------------------------------------- #
# dataset loading (host side)
# --------------------------------------------------------------------------- #
DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data')

all_labels = None


def sample_mask(idx, l):
    """Create a boolean mask (gcn/utils.py:42-46)."""
    mask = np.zeros(l)
    mask[idx] = 1
    return np.array(mask, dtype=bool)


def adjacency_from_dict_of_lists(keys, ptr, nbr):
    """What ``nx.adjacency_matrix(nx.from_dict_of_lists(graph))`` yields (gcn/utils.py:123): an
    undirected simple graph over the nodes in first-seen order, every edge stored in both
    directions with value 1, a self-loop once."""
    order = {}
    for k in keys.tolist():
        order.setdefault(k, len(order))
    for v in nbr.tolist():
        order.setdefault(v, len(order))
    n = len(order)
    src = np.repeat(np.asarray([order[k] for k in keys.tolist()], dtype=np.int64), np.diff(ptr))
    dst = np.asarray([order[v] for v in nbr.tolist()], dtype=np.int64)
    rows = np.concatenate([src, dst])
    cols = np.concatenate([dst, src])
    key = np.unique(rows * n + cols)
    return sp.csr_matrix((np.ones(key.size, dtype=np.int64), (key // n, key % n)), shape=(n, n))


def preprocess_features(features, feature_type):
    """gcn/utils.py:278-294.  'bow': row-normalise (r_inv = rowsum^-1, inf -> 0); 'none': identity
    features.  'tfidf' needs scikit-learn's transformer and is not part of this build."""
    if feature_type == 'bow':
        rowsum = np.array(features.sum(1))
        with np.errstate(divide='ignore'):
            r_inv = np.power(rowsum, -1).flatten()
        r_inv[np.isinf(r_inv)] = 0.
        features = sp.diags(r_inv).dot(features)
    elif feature_type == 'none':
        features = sp.csr_matrix(sp.eye(features.shape[0]))
    else:
        raise ValueError('Invalid feature type: ' + str(feature_type))
    return features


def load_data(dataset_str, train_size, validation_size, model_config, shuffle=True, data_dir=None):
    """Load a Planetoid dataset and draw the reference's random split (gcn/utils.py:102-255).

    The files are the reference's ``ind.<dataset>.*`` contents re-encoded as arrays in
    ``data/<dataset>_raw.npz``.  Returns ``adj, features, y_train, y_val, y_test, train_mask,
    val_mask, test_mask`` exactly as the reference does; uses numpy's global RandomState for the
    shuffle, so ``np.random.seed(seed)`` beforehand reproduces the reference's split."""
    global all_labels
    path = os.path.join(data_dir or DATA_DIR, dataset_str + '_raw.npz')
    if not os.path.exists(path):
        raise ValueError('dataset "%s" is not available (%s missing)' % (dataset_str, path))
    raw = np.load(path)

    def csr(part):
        return sp.csr_matrix((raw[part + '_data'], raw[part + '_indices'], raw[part + '_indptr']),
                             shape=tuple(raw[part + '_shape']))

    x, tx, allx = csr('x'), csr('tx'), csr('allx')
    y, ty, ally = raw['y'].astype(np.int32), raw['ty'].astype(np.int32), raw['ally'].astype(np.int32)
    adj = adjacency_from_dict_of_lists(raw['graph_keys'], raw['graph_ptr'], raw['graph_nbr'])
    test_idx_reorder = raw['test_index'].tolist()
    test_idx_range = np.sort(test_idx_reorder)

    if dataset_str == 'citeseer':
        # isolated test nodes become zero rows at the right position (utils.py:127-136)
        full = range(min(test_idx_reorder), max(test_idx_reorder) + 1)
        tx_extended = sp.lil_matrix((len(full), x.shape[1]))
        tx_extended[test_idx_range - min(test_idx_range), :] = tx
        tx = tx_extended
        ty_extended = np.zeros((len(full), y.shape[1]))
        ty_extended[test_idx_range - min(test_idx_range), :] = ty
        ty = ty_extended

    features = sp.vstack((allx, tx)).tolil()
    labels = np.vstack((ally, ty))
    features[test_idx_reorder, :] = features[test_idx_range, :]
    labels[test_idx_reorder, :] = labels[test_idx_range, :]
    features = preprocess_features(features, feature_type=model_config['feature'])
    all_labels = labels.copy()

    # split (utils.py:183-215): walk a shuffled order, take nodes until every class has train_size
    idx = np.arange(len(labels))
    no_class = labels.shape[1]
    label_each_class = [train_size for _ in range(no_class)]
    if shuffle:
        np.random.shuffle(idx)
    idx_train = []
    count = [0 for _ in range(no_class)]
    nxt = 0
    for i in idx:
        if count == label_each_class:
            break
        nxt += 1
        for j in range(no_class):
            if labels[i, j] and count[j] < label_each_class[j]:
                idx_train.append(i)
                count[j] += 1

    test_size = model_config['test_size']
    if model_config['validate']:
        if test_size:
            assert nxt + validation_size < len(idx)
        idx_val = idx[nxt:nxt + validation_size]
        assert nxt + validation_size + (test_size or 0) < len(idx)
        idx_test = idx[-test_size:] if test_size else idx[nxt + validation_size:]
    else:
        if test_size:
            assert nxt + test_size < len(idx)
        idx_val = idx[-test_size:] if test_size else idx[nxt:]
        idx_test = idx[-test_size:] if test_size else idx[nxt:]
    print('labels of each class : ', np.sum(labels[idx_train], axis=0))

    train_mask = sample_mask(idx_train, labels.shape[0])
    val_mask = sample_mask(idx_val, labels.shape[0])
    test_mask = sample_mask(idx_test, labels.shape[0])
    y_train = np.zeros(labels.shape)
    y_val = np.zeros(labels.shape)
    y_test = np.zeros(labels.shape)
    y_train[train_mask, :] = labels[train_mask, :]
    y_val[val_mask, :] = labels[val_mask, :]
    y_test[test_mask, :] = labels[test_mask, :]
    return adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask
