"""CPU: the build's own loader (gcn_amd/utils.py load_data, preprocess_features, sparse_to_tuple)
against what the reference's load_data returned for the same seed (tests/golden/*.npz)."""
import numpy as np
import scipy.sparse as sp

MODEL_CONFIG = {'feature': 'bow', 'test_size': None, 'validate': False}


def test_load_data_matches_reference(dataset_fx):
    from gcn_amd import utils
    name, fx = dataset_fx
    np.random.seed(int(fx['split_seed']))
    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = utils.load_data(
        name, train_size=int(fx['train_size']), validation_size=500, model_config=dict(MODEL_CONFIG), shuffle=True)
    # graph: same nodes, same undirected edges, values 1
    coo = sp.coo_matrix(adj)
    order = np.lexsort((coo.col, coo.row))
    assert np.array_equal(coo.row[order], fx['adj_row']) and np.array_equal(coo.col[order], fx['adj_col'])
    assert np.all(coo.data == 1)
    # features: bag-of-words rows, identical bits and identical storage order
    feats = sp.csr_matrix(features)
    assert np.array_equal(feats.indptr, fx['feat_indptr']) and np.array_equal(feats.indices, fx['feat_indices'])
    assert feats.data.dtype == fx['feat_data'].dtype and np.array_equal(feats.data, fx['feat_data'])
    triple = utils.sparse_to_tuple(features)
    assert np.array_equal(triple.indices, fx['feat_coords']) and np.array_equal(triple.values, fx['feat_values'])
    assert np.array_equal(triple.dense_shape, fx['feat_shape'])
    # labels and the random split
    assert np.array_equal(utils.all_labels, fx['labels'])
    for got, key in [(y_train, 'y_train'), (y_val, 'y_val'), (y_test, 'y_test'), (train_mask, 'train_mask'),
                     (val_mask, 'val_mask'), (test_mask, 'test_mask')]:
        assert np.array_equal(got, fx[key]), key
    assert int(train_mask.sum()) == int(fx['train_size']) * y_train.shape[1]
    # validate=False: validation set == test set == everything after the scan cursor (utils.py:214-215)
    assert np.array_equal(val_mask, test_mask)


def test_unknown_inputs_raise_value_error():
    import pytest
    from gcn_amd import utils
    with pytest.raises(ValueError):
        utils.load_data('pubmed', 20, 500, dict(MODEL_CONFIG))        # allx is absent upstream
    with pytest.raises(ValueError):
        utils.preprocess_features(sp.eye(3).tocsr(), 'tfidf')
    cfg = {'Model': 'GCN', 'connection': 'cx', 'layer_size': [16], 'conv': 'gcn', 'name': None}
    with pytest.raises(ValueError):
        utils.preprocess_model_config(cfg)
    cfg = {'Model': 'GCN', 'connection': 'cc', 'layer_size': [16, 8], 'conv': 'gcn', 'name': None}
    with pytest.raises(ValueError):
        utils.preprocess_model_config(cfg)
    cfg = {'Model': 'GCN', 'connection': 'cc', 'layer_size': [16], 'conv': 'gcn', 'name': None, 'validate': False}
    utils.preprocess_model_config(cfg)
    assert cfg['name'] == 'GCN_c16c' and cfg['connection'] == ['c', 'c']
