"""config.py keeps the reference's keys, defaults and flags (config.py:10-64, 92-107 of the reference;
the expected values below are that file's settings written out as data)."""
import os
import sys

from conftest import ROOT

EXPECTED_DEFAULTS = {
    'dataset': 'cora', 'shuffle': True, 'train_size': 2, 'validation_size': 500, 'validate': False,
    'test_size': None, 'conv': 'gcn', 'max_degree': 2, 'learning_rate': 0.02, 'epochs': 200, 'Model': 0,
    'Model_to_add_label': {'Model': 0}, 'Model_to_predict': {'Model': 0}, 'Model19': 'union', 'alpha': 1e-6,
    'connection': 'cc', 'layer_size': [16], 'dropout': 0.5, 'weight_decay': 5e-4, 'feature': 'bow',
    'model_dir': './model/', 'name': None, 'train': True,
}


def _config():
    sys.path.insert(0, ROOT)
    import config
    return config


def test_defaults_are_the_reference_defaults():
    d = _config().default_model_config()
    for key, value in EXPECTED_DEFAULTS.items():
        assert d[key] == value, key
    assert set(d) == set(EXPECTED_DEFAULTS) | {'random_seed', 'threads'}
    assert d['threads'] == 2 * os.cpu_count() and isinstance(d['random_seed'], int)


def test_flags_override_as_in_the_reference():
    config = _config()
    conf, args = config.load_configuration(['--dataset', 'citeseer', '--train_size', '20', '--repeating', '3', '-v'],
                                           echo=False)
    assert conf['repeating'] == 3 and args.verbose
    model, = conf['model_list']
    assert model['Model'] == 'GCN' and model['dataset'] == 'citeseer' and model['train_size'] == 20
    # a list gives one count per class (config.py:19, 105)
    conf, _ = config.load_configuration(['--train_size', '[5, 5, 5, 5, 5, 5, 5]'], echo=False)
    assert conf['model_list'][0]['train_size'] == [5] * 7
    # the name is generated from the architecture when none is given (gcn/utils.py preprocess_model_config)
    assert conf['model_list'][0]['name']


def test_values_outside_the_build_are_refused_when_the_configuration_is_read():
    import pytest
    import config
    cfg = config.default_model_config()
    cfg.update({'Model': 'GCN', 'connection': ['c', 'c']})
    config.check_supported(cfg)
    for key, value in (('Model', 'cotraining'), ('Model', 'lp'), ('conv', 'cheby'), ('feature', 'tfidf')):
        bad = dict(cfg)
        bad[key] = value
        with pytest.raises(ValueError, match='outside this build'):
            config.check_supported(bad)
    bad = dict(cfg)
    bad['connection'] = ['c', 'f']
    with pytest.raises(ValueError, match='connection letter'):
        config.check_supported(bad)
