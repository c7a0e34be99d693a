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


HUB_COUNT = 65536          # vertices moved to the front by reorder='hubs'


def vertex_order(indptr, reorder):
    """(new_id, old_id) of a vertex order named by ``reorder``: 'degree' (descending degree), 'hubs' (the
    HUB_COUNT vertices of highest degree first, the others in their old order) or ('hubs', K)."""
    if reorder == 'degree':
        return sparse.degree_order(indptr)
    if reorder == 'hubs':
        return sparse.hub_order(indptr, HUB_COUNT)
    if isinstance(reorder, (tuple, list)) and len(reorder) == 2 and reorder[0] == 'hubs':
        return sparse.hub_order(indptr, int(reorder[1]))
    raise ValueError("reorder must be None, 'degree', 'hubs' or ('hubs', K), not %r" % (reorder,))


def relabelled_csr(csr, new_id):
    """P . A . P^T of a square float32 DeviceCSR for the labels ``new_id``: columns ascend inside every row
    of the result, values move with their entries."""
    n = csr.n_rows
    ip, ix, v = sparse.csr_relabel(n, n, csr.indptr, csr.indices, csr.values.double(), new_id, new_id)
    return sparse.DeviceCSR(n, n, ip, ix.contiguous(), v.to(torch.float32))


def preprocess_adj(adj, type='sym', loop=True, device='cuda', reorder=None):
    """Preprocessing of the adjacency matrix for the GCN model and conversion to the triple
    representation (gcn/utils.py:318-323): ``adj + I`` (ADDS 1 to the diagonal), degree^-1/2
    scaling in float64, triple in the element order of the reference (column-major for 'sym').
    The returned triple's ``.csr`` is the float32 device CSR of the same matrix.

    ``reorder`` ('degree', 'hubs' or ('hubs', K); the reference has no such step): the device CSR is the
    matrix of the graph with its vertices relabelled once, by an integer kernel, so that the rows of the dense
    operand that most entries read sit next to each other; the triple itself stays in the dataset's order, and
    ``.new_id`` / ``.old_id`` (int32 device tensors) carry the permutation.  A model fed this support
    permutes features, labels and masks in and its outputs back (GCN_MLP does), so callers see the dataset's
    order throughout; products then add a row's terms in the order of the NEW column numbers."""
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
    out.new_id = out.old_id = None
    if reorder is not None:
        out.new_id, out.old_id = vertex_order(csr.indptr, reorder)
        csr = relabelled_csr(csr, out.new_id)
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
    """Feature preprocessing of gcn/utils.py:278-294: 'bow' scales every row by the reciprocal of its sum
    (a row that sums to zero stays zero); 'none' replaces the features by the identity.  'tfidf' needs
    scikit-learn's transformer and is not part of this build."""
    if feature_type == 'none':
        return sp.csr_matrix(sp.eye(features.shape[0]))
    if feature_type != 'bow':
        raise ValueError('Invalid feature type: ' + str(feature_type))
    totals = np.asarray(features.sum(axis=1)).ravel()
    with np.errstate(divide='ignore'):
        scale = np.where(totals != 0, np.power(totals, -1.0), 0.0)
    return sp.diags(scale).dot(features)


def _take_rows(matrix, source_rows):
    """Row gather of a CSR matrix: row i of the result is row source_rows[i] of ``matrix``."""
    return sp.csr_matrix(matrix)[source_rows]


def _spread_rows(block, positions, n_rows):
    """A float64 matrix of ``n_rows`` rows holding row k of ``block`` at row positions[k], zeros elsewhere
    (the reference pads into a default-typed, i.e. float64, matrix: CiteSeer's features come out float64)."""
    coo = sp.coo_matrix(block)
    return sp.csr_matrix((coo.data.astype(np.float64), (positions[coo.row], coo.col)),
                         shape=(n_rows, block.shape[1]))


class Split(object):
    """The reference's random split (gcn/utils.py:183-237) computed from positions in the shuffled order.

    The reference walks the shuffled node order and takes a node for training while its class still
    needs members, until every class holds ``per_class``; the walk's end is the cursor after which
    validation and test nodes are taken.  Walking is not needed: a node is taken exactly when it is among
    the first ``per_class`` of its class in that order, and the cursor sits just behind the node that
    completes the last class (at the end when some class never fills up)."""

    def __init__(self, labels, order, per_class):
        member = labels[order] != 0                                  # [position, class]
        seen = np.cumsum(member, axis=0)                             # members of each class up to a position
        taken_pos, taken_cls = np.nonzero(member & (seen <= per_class))
        self.train = order[taken_pos]                                # position order; a multi-label row once per class
        if per_class <= 0 or member.shape[0] == 0:
            self.cursor = 0
        else:
            complete = np.all(seen >= per_class, axis=1)
            self.cursor = int(np.argmax(complete)) + 1 if complete.any() else len(order)
        self.order = order

    def validation_and_test(self, validate, validation_size, test_size):
        order, cursor, n = self.order, self.cursor, len(self.order)
        tail = order[n - test_size:] if test_size else None
        if validate:
            assert cursor + validation_size + (test_size or 0) < n, 'not enough nodes left for validation and test'
            return order[cursor:cursor + validation_size], (tail if test_size else order[cursor + validation_size:])
        if test_size:
            assert cursor + test_size < n, 'not enough nodes left for the test set'
            return tail, tail
        return order[cursor:], order[cursor:]


def load_data(dataset_str, train_size, validation_size, model_config, shuffle=True, data_dir=None):
    """Load a Planetoid dataset and draw the reference's random split (gcn/utils.py:102-255).

    The files are the reference's ``ind.<dataset>.*`` contents re-encoded as arrays in
    ``data/<dataset>_raw.npz``.  Returns ``adj, features, y_train, y_val, y_test, train_mask,
    val_mask, test_mask`` as the reference does.  The node order is shuffled with numpy's global
    RandomState (one ``np.random.shuffle`` of ``arange(n)``, the reference's only draw), so
    ``np.random.seed(seed)`` beforehand reproduces the reference's split for that seed."""
    global all_labels
    path = os.path.join(data_dir or DATA_DIR, dataset_str + '_raw.npz')
    if not os.path.exists(path):
        raise ValueError('dataset "%s" is not available (%s missing)' % (dataset_str, path))
    raw = np.load(path)

    def part(name):
        return sp.csr_matrix((raw[name + '_data'], raw[name + '_indices'], raw[name + '_indptr']),
                             shape=tuple(raw[name + '_shape']))

    adj = adjacency_from_dict_of_lists(raw['graph_keys'], raw['graph_ptr'], raw['graph_nbr'])
    test_ids = raw['test_index'].astype(np.int64)                    # node ids of the test rows, in file order
    test_sorted = np.sort(test_ids)
    test_x, test_y = part('tx'), raw['ty'].astype(np.int32)
    span = int(test_sorted[-1] - test_sorted[0]) + 1
    if span != test_x.shape[0]:
        # ids missing from the test range (CiteSeer's isolated nodes, utils.py:127-136) become zero rows
        at = test_sorted - test_sorted[0]
        test_x = _spread_rows(test_x, at, span)
        padded = np.zeros((span, test_y.shape[1]))
        padded[at] = test_y
        test_y = padded
    stacked_x = sp.vstack((part('allx'), test_x)).tocsr()
    stacked_y = np.vstack((raw['ally'].astype(np.int32), test_y))
    # the test rows are stored in file order: move each to the row of its node id (utils.py:175-176)
    source = np.arange(stacked_y.shape[0])
    source[test_ids] = test_sorted
    labels = stacked_y[source]
    features = preprocess_features(_take_rows(stacked_x, source).tolil(), feature_type=model_config['feature'])
    all_labels = labels.copy()

    order = np.arange(labels.shape[0])
    if shuffle:
        np.random.shuffle(order)
    split = Split(labels, order, train_size)
    held_out_val, held_out_test = split.validation_and_test(model_config['validate'], validation_size,
                                                            model_config['test_size'])
    print('labels of each class : ', np.sum(labels[split.train], axis=0))

    def masked(nodes):
        mask = sample_mask(nodes, labels.shape[0])
        y = np.zeros(labels.shape)
        y[mask] = labels[mask]
        return y, mask

    (y_train, train_mask), (y_val, val_mask), (y_test, test_mask) = (masked(split.train), masked(held_out_val),
                                                                     masked(held_out_test))
    return adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask
