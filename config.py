"""Run configuration with the reference's keys, defaults and command-line flags (config.py:10-120).

The reference parses ``sys.argv`` while it is being imported; here parsing happens in
``load_configuration(argv)`` so the module can be imported by tests, and ``python train.py``
calls it with the process arguments.  Only ``{'Model': 'GCN'}`` is in this build's model list:
the label-expansion models of the reference are outside the propagation path.
"""
import argparse
import pprint
import time
from copy import deepcopy
from os import cpu_count

from gcn_amd.utils import preprocess_model_config


def default_model_config():
    d = {}
    d['dataset'] = 'cora'              # cora | citeseer (pubmed's feature file is absent upstream)
    d['shuffle'] = True
    d['train_size'] = 2                # labels per class (a list gives one count per class)
    d['validation_size'] = 500
    d['validate'] = False              # no validation set: test on best TRAIN accuracy
    d['test_size'] = None              # None: all remaining nodes are the test set
    d['conv'] = 'gcn'                  # gcn | gcn_rw | gcn_unnorm | gcn_noloop
    d['max_degree'] = 2
    d['learning_rate'] = 0.02
    d['epochs'] = 200
    d['Model'] = 0
    d['Model_to_add_label'] = {'Model': 0}
    d['Model_to_predict'] = {'Model': 0}
    d['Model19'] = 'union'
    d['alpha'] = 1e-6
    d['connection'] = 'cc'             # one letter per layer; "c" = graph convolution
    d['layer_size'] = [16]             # hidden widths; len(connection) == len(layer_size) + 1
    d['dropout'] = 0.5
    d['weight_decay'] = 5e-4
    d['random_seed'] = int(time.time())
    d['feature'] = 'bow'               # bow | none
    d['model_dir'] = './model/'
    d['name'] = None                   # generated from the architecture when empty
    d['threads'] = 2 * cpu_count()
    d['train'] = True
    return d


# What of the reference's configuration space this build runs: the GCN propagation path only.
SUPPORTED = {
    'Model': ('GCN',),                                              # not: cotraining, selftraining, lp, union, intersection
    'conv': ('gcn', 'gcn_rw', 'gcn_unnorm', 'gcn_noloop'),          # not: cheby, chebytheta
    'feature': ('bow', 'none'),                                     # not: tfidf
    'connection': ('c',),                                           # not: f (dense), d (DenseNet), r (residual), C
}


def check_supported(model_config):
    """Refuse, while the configuration is being read, what the reference accepts and this build does not
    run -- instead of failing later inside train()."""
    for key in ('Model', 'conv', 'feature'):
        if model_config[key] not in SUPPORTED[key]:
            raise ValueError('%s = %r is outside this build (the GCN propagation path): supported values are %s'
                             % (key, model_config[key], ', '.join(map(repr, SUPPORTED[key]))))
    for letter in model_config['connection']:
        if letter not in SUPPORTED['connection']:
            raise ValueError('connection letter %r is outside this build: only graph-convolution layers ("c") run here'
                             % letter)


def build_parser():
    parser = argparse.ArgumentParser(description=(
        'Train and test a Graph Convolution Network for node classification on the MI355X path.\n'
        'Most settings live in config.py.'))
    parser.add_argument('-v', '--verbose', action='store_true')
    parser.add_argument('--dataset', type=str)
    parser.add_argument('--train_size', type=str)
    parser.add_argument('--repeating', type=int)
    parser.add_argument('--hip_graph', action='store_true',
                        help='record the train step once as a HIP graph and replay it (not a flag of the reference)')
    return parser


def load_configuration(argv=None, echo=True):
    """Returns ``(configuration, args)`` shaped like the reference's module globals."""
    args = build_parser().parse_args(argv)
    configuration = {'repeating': 1, 'default': default_model_config(), 'model_list': [{'Model': 'GCN'}]}
    if args.dataset is not None:
        configuration['default']['dataset'] = args.dataset
    if args.train_size is not None:
        # the reference evaluates this string (config.py:105) so that "20" and "[20,20,...]" both work
        import ast
        configuration['default']['train_size'] = ast.literal_eval(args.train_size)
    if args.repeating is not None:
        configuration['repeating'] = args.repeating
    if args.hip_graph:
        configuration['default']['hip_graph'] = True
    if echo:
        print(args)
        pprint.PrettyPrinter(indent=4).pprint(configuration)
    merged = []
    for model in configuration['model_list']:
        model_config = deepcopy(configuration['default'])
        model_config.update(model)
        preprocess_model_config(model_config)
        check_supported(model_config)
        merged.append(model_config)
    configuration['model_list'] = merged
    return configuration, args
