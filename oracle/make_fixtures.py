#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference's own preprocessing.

TEST INFRASTRUCTURE ONLY; runs in the development container, where
``/root/reference`` exists.  Nothing here is imported by the product.

What runs from the reference (imported, never copied): ``gcn/utils.py``
``load_data`` (102-255), ``preprocess_features`` (278-294), ``normalize_adj``
(297-315), ``preprocess_adj`` (318-323), ``sparse_to_tuple`` (258-275).

Two modules the file imports do not exist here and are never used on this
path, so empty stand-ins are registered before the import: ``tensorflow``
(only ``tf.SparseTensorValue``, a namedtuple, is touched: utils.py:267) and
``scipy.sparse.linalg.eigen.arpack`` (``eigsh`` is imported at utils.py:10 and
never called).  The reference opens its data files with ``pickle``; that name
is rebound to the data-only reader in ``planetoid_reader.py`` so no pickle is
ever executed.

Usage:  python oracle/make_fixtures.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import collections
import importlib
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from planetoid_reader import read_planetoid_file  # noqa: E402

SPLIT_SEED = 123
TRAIN_SIZE = 20
MODEL_CONFIG = {'feature': 'bow', 'test_size': None, 'validate': False}


def import_reference_utils(ref_root):
    tf = types.ModuleType('tensorflow')
    tf.SparseTensorValue = collections.namedtuple('SparseTensorValue', ['indices', 'values', 'dense_shape'])
    sys.modules['tensorflow'] = tf
    eigen = types.ModuleType('scipy.sparse.linalg.eigen')
    arpack = types.ModuleType('scipy.sparse.linalg.eigen.arpack')
    arpack.eigsh = None
    eigen.arpack = arpack
    sys.modules['scipy.sparse.linalg.eigen'] = eigen
    sys.modules['scipy.sparse.linalg.eigen.arpack'] = arpack
    sys.path.insert(0, ref_root)
    sys.dont_write_bytecode = True
    utils = importlib.import_module('gcn.utils')
    # the reference calls pkl.load(f, encoding='latin1'): route it to the data-only reader
    utils.pkl = types.SimpleNamespace(load=lambda f, **kw: read_planetoid_file(f.name))
    return utils


def sorted_coo(adj):
    coo = sp.coo_matrix(adj)
    order = np.lexsort((coo.col, coo.row))
    return coo.row[order].astype(np.int32), coo.col[order].astype(np.int32), coo.data[order]


def graph_fixture(utils, adj):
    """Everything pinned about preprocess_adj for one graph."""
    out = {}
    r, c, v = sorted_coo(adj)
    assert np.all(v == 1)
    out['adj_row'], out['adj_col'] = r, c
    out['n'] = np.int64(adj.shape[0])
    hat = utils.preprocess_adj(adj)                       # utils.py:318-323 as called at train.py:125
    out['hat_indices'] = np.asarray(hat.indices, dtype=np.int32)
    assert np.array_equal(out['hat_indices'], hat.indices)
    out['hat_values'] = np.asarray(hat.values, dtype=np.float64)
    out['hat_shape'] = np.asarray(hat.dense_shape, dtype=np.int64)
    # derived integer targets: CSR of A+I and its row sums
    ai = sp.csr_matrix(adj + sp.eye(adj.shape[0]))
    ai.sum_duplicates()
    ai.sort_indices()
    out['ai_indptr'] = ai.indptr.astype(np.int32)
    out['ai_indices'] = ai.indices.astype(np.int32)
    out['ai_data'] = ai.data.astype(np.float64)
    out['deg'] = np.asarray(ai.sum(1)).ravel().astype(np.float64)
    # sibling supports (train.py:130-135)
    noloop = utils.preprocess_adj(adj, loop=False)
    out['noloop_indices'] = np.asarray(noloop.indices, dtype=np.int32)
    out['noloop_values'] = np.asarray(noloop.values, dtype=np.float64)
    rw = utils.preprocess_adj(adj, type='rw')
    out['rw_indices'] = np.asarray(rw.indices, dtype=np.int32)
    out['rw_values'] = np.asarray(rw.values, dtype=np.float64)
    return out


def dataset_fixture(utils, name):
    np.random.seed(SPLIT_SEED)
    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = utils.load_data(
        name, train_size=TRAIN_SIZE, validation_size=500, model_config=dict(MODEL_CONFIG), shuffle=True)
    out = graph_fixture(utils, adj)
    feats = sp.csr_matrix(features)
    out['feat_indptr'] = feats.indptr.astype(np.int32)
    out['feat_indices'] = feats.indices.astype(np.int32)
    out['feat_data'] = feats.data
    triple = utils.sparse_to_tuple(features)              # train.py:118
    out['feat_coords'] = np.asarray(triple.indices, dtype=np.int32)
    out['feat_values'] = np.asarray(triple.values)
    out['feat_shape'] = np.asarray(triple.dense_shape, dtype=np.int64)
    out['labels'] = utils.all_labels.astype(np.int8)
    out['y_train'] = y_train.astype(np.int8)
    out['y_val'] = y_val.astype(np.int8)
    out['y_test'] = y_test.astype(np.int8)
    out['train_mask'] = np.asarray(train_mask, dtype=np.bool_)
    out['val_mask'] = np.asarray(val_mask, dtype=np.bool_)
    out['test_mask'] = np.asarray(test_mask, dtype=np.bool_)
    out['split_seed'] = np.int64(SPLIT_SEED)
    out['train_size'] = np.int64(TRAIN_SIZE)
    return out


def raw_dataset(ref_root, name):
    """Raw file contents re-encoded as arrays, so the build's own loader can be
    tested (and run on the GPU box) without the reference tree."""
    out = {}
    for part in ['x', 'tx', 'allx']:
        p = os.path.join(ref_root, 'data', 'ind.%s.%s' % (name, part))
        if not os.path.exists(p):
            continue
        m = read_planetoid_file(p)
        out[part + '_indptr'] = m.indptr.astype(np.int32)
        out[part + '_indices'] = m.indices.astype(np.int32)
        out[part + '_data'] = m.data
        out[part + '_shape'] = np.asarray(m.shape, dtype=np.int64)
    for part in ['y', 'ty', 'ally']:
        out[part] = read_planetoid_file(os.path.join(ref_root, 'data', 'ind.%s.%s' % (name, part))).astype(np.int8)
    graph = read_planetoid_file(os.path.join(ref_root, 'data', 'ind.%s.graph' % name))
    keys = list(graph.keys())                             # insertion order matters to networkx node numbering
    out['graph_keys'] = np.asarray(keys, dtype=np.int32)
    lens = np.asarray([len(graph[k]) for k in keys], dtype=np.int64)
    out['graph_ptr'] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out['graph_nbr'] = np.asarray([v for k in keys for v in graph[k]], dtype=np.int32)
    with open(os.path.join(ref_root, 'data', 'ind.%s.test.index' % name)) as f:
        out['test_index'] = np.asarray([int(line.strip()) for line in f], dtype=np.int32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    utils = import_reference_utils(args.ref)
    os.chdir(args.ref)                                    # data paths are relative (utils.py:116)
    import networkx as nx

    for name in ['cora', 'citeseer']:
        fx = dataset_fixture(utils, name)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), **fx)
        np.savez_compressed(os.path.join(os.path.dirname(out_dir), '..', 'data', name + '_raw.npz'),
                            **raw_dataset(args.ref, name))
        print(name, 'n', int(fx['n']), 'nnz(A)', fx['adj_row'].size, 'nnz(hat)', fx['hat_values'].size,
              'nnz(X)', fx['feat_values'].size, fx['feat_values'].dtype)

    # PubMed: allx is absent upstream, so only the graph is pinned (utils.py:123)
    graph = read_planetoid_file(os.path.join(args.ref, 'data', 'ind.pubmed.graph'))
    adj = nx.adjacency_matrix(nx.from_dict_of_lists(graph))
    fx = graph_fixture(utils, adj)
    np.savez_compressed(os.path.join(out_dir, 'pubmed_graph.npz'), **fx)
    print('pubmed', 'n', int(fx['n']), 'nnz(A)', fx['adj_row'].size, 'nnz(hat)', fx['hat_values'].size)


if __name__ == '__main__':
    main()
