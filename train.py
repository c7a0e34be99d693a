"""Training driver with the reference's loop (train.py:15-306) for ``Model == 'GCN'``.

    python train.py [-v] [--dataset cora|citeseer] [--train_size K] [--repeating R]

``train(model_config, sess, seed, data_split=None)`` keeps the reference's signature and
return value ``(test_acc, test_acc_of_class, prediction, seconds)``.  Per epoch it runs one
optimiser step (the timed region of train.py:203-205), one more train-mode evaluation
(train.py:207) and, when train accuracy strictly improves, a test evaluation and a
prediction (train.py:227-234).
"""
from __future__ import division, print_function

import time

import numpy as np
from scipy import sparse

from gcn_amd import graph as tf
from gcn_amd.models import GCN_MLP
from gcn_amd.utils import construct_feed_dict, load_data, preprocess_adj, sparse_to_tuple

args = None


def _sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def build_support(adj, conv):
    """The support list per convolution type (train.py:124-141)."""
    if conv == 'gcn':
        return [preprocess_adj(adj)]
    if conv == 'gcn_unnorm':
        return [sparse_to_tuple(adj.astype(np.float32))]
    if conv == 'gcn_noloop':
        return [preprocess_adj(adj, loop=False)]
    if conv == 'gcn_rw':
        return [preprocess_adj(adj, type='rw')]
    raise ValueError('Invalid argument for model_config["conv"]: ' + str(conv))


class Evaluation(object):
    """Loss, accuracy and per-class accuracy of one feed, with the wall time the fetch took."""

    def __init__(self, sess, model, feed, with_prediction=False):
        fetches = [model.loss, model.accuracy, model.accuracy_of_class] + ([model.prediction] if with_prediction else [])
        started = time.time()
        values = sess.run(fetches, feed_dict=feed)
        self.seconds = time.time() - started
        self.cost, self.accuracy, self.accuracy_of_class = values[:3]
        self.prediction = values[3] if with_prediction else None


def _inputs(model_config, data_split):
    """adj, features, labels and masks: from the caller's split or drawn by load_data (train.py:26-40)."""
    names = ('adj', 'features', 'y_train', 'y_val', 'y_test', 'train_mask', 'val_mask', 'test_mask')
    if data_split:
        return tuple(data_split[k] for k in names)
    return load_data(model_config['dataset'], train_size=model_config['train_size'],
                     validation_size=model_config['validation_size'], model_config=model_config,
                     shuffle=model_config['shuffle'])


def _placeholders(n_supports, n_classes):
    """The feed slots of train.py:144-154."""
    return {
        'support': [tf.sparse_placeholder(tf.float32, name='support' + str(i)) for i in range(n_supports)],
        'features': tf.sparse_placeholder(tf.float32, name='features'),
        'labels': tf.placeholder(tf.int32, name='labels', shape=(None, n_classes)),
        'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
        'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
    }


def _timed_step(sess, model, feed):
    """One optimiser step between two device synchronisations: the region train.py:203-205 times."""
    _sync()
    started = time.time()
    sess.run(model.opt_op, feed_dict=feed)
    _sync()
    return time.time() - started


def train(model_config, sess, seed, data_split=None):
    """The reference's train() (train.py:15-257) for Model == 'GCN'.  Every epoch: one timed optimiser
    step, one more evaluation of the training feed with dropout on (train.py:207), and -- when the
    monitored accuracy (training, or validation with ``validate``) strictly improves -- a test evaluation
    and a prediction on the training feed (train.py:227-234).  The reported test result is therefore the
    one of the last improving epoch, not of the final one."""
    verbose = bool(args is not None and args.verbose)
    began = time.time()
    print('', *['{:<15}: {}'.format(key, model_config[key]) for key in
                ('name', 'dataset', 'train_size', 'learning_rate', 'feature')], sep='\n')
    if model_config['Model'] != 'GCN':
        raise ValueError("model_config['Model'] must be 'GCN' in this build, but is {} now".format(
            model_config['Model']))

    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = _inputs(model_config, data_split)
    if not sparse.issparse(features):
        raise ValueError('this build feeds bag-of-words features as a sparse triple')
    features = sparse_to_tuple(features)
    support = build_support(adj, model_config['conv'])
    placeholders = _placeholders(len(support), y_train.shape[1])
    model = GCN_MLP(model_config, placeholders, input_dim=features[2][1])
    sess.run(tf.global_variables_initializer())

    feeds = {which: construct_feed_dict(features, support, y, mask, placeholders)
             for which, y, mask in (('train', y_train, train_mask), ('valid', y_val, val_mask),
                                    ('test', y_test, test_mask))}
    feeds['train'][placeholders['dropout']] = model_config['dropout']

    test = Evaluation(sess, model, feeds['test'], with_prediction=True)    # before any training (train.py:192)
    prediction = test.prediction
    monitored = 'valid' if model_config['validate'] else 'train'
    best = 0
    step_seconds = []
    if model_config['train']:
        print('training...')
        for epoch in range(model_config['epochs']):
            step_seconds.append(_timed_step(sess, model, feeds['train']))
            seen = {'train': sess.run([model.loss, model.accuracy, model.summary], feed_dict=feeds['train'])[:2]}
            if model_config['validate']:
                seen['valid'] = sess.run([model.loss, model.accuracy, model.summary], feed_dict=feeds['valid'])[:2]
            improved = seen[monitored][1] > best
            if improved:
                best = seen[monitored][1]
                test = Evaluation(sess, model, feeds['test'])
                prediction = sess.run(model.prediction, feeds['train'])
            if verbose:
                parts = ['*' if improved else '', 'Epoch: {:04d}'.format(epoch),
                         'train_loss= {:.3f}'.format(seen['train'][0]), 'train_acc= {:.3f}'.format(seen['train'][1])]
                if 'valid' in seen:
                    parts += ['val_loss= {:.3f}'.format(seen['valid'][0]), 'val_acc= {:.3f}'.format(seen['valid'][1])]
                print(parts[0] + ' '.join(parts[1:]), 'time= {:.5f}'.format(step_seconds[-1]))
        print('Optimization Finished!')
        print('Test set results:', 'cost=', '{:.5f}'.format(test.cost), 'accuracy=', '{:.5f}'.format(test.accuracy),
              'time=', '{:.5f}'.format(test.seconds))
        print('accuracy of each class=', test.accuracy_of_class)
        if step_seconds:
            print('epoch ms (optimiser step): median {:.3f}  min {:.3f}'.format(
                1e3 * float(np.median(step_seconds)), 1e3 * float(np.min(step_seconds))))
    total = time.time() - began
    print('Total time={}s'.format(total))
    return test.accuracy, test.accuracy_of_class, prediction, total


class Tally(object):
    """Results of the repetitions of one model: what the closing table of train.py:284-306 averages."""

    def __init__(self, model_config):
        self.model_config = model_config
        self.accuracy, self.accuracy_of_class, self.seconds = [], [], []

    def add(self, accuracy, accuracy_of_class, seconds):
        self.accuracy.append(accuracy)
        self.accuracy_of_class.append(accuracy_of_class)
        self.seconds.append(seconds)

    def row(self):
        cfg = self.model_config
        cells = (cfg['dataset'], str(cfg['train_size']) + ' per class', str(cfg['validation_size']))
        return '\t'.join(['{:<8}'.format(c) for c in cells] +
                         ['{:<8.6f}'.format(np.mean(self.accuracy)), '{:<8.6f}'.format(np.std(self.accuracy)),
                          '{:<8.2f}'.format(np.mean(self.seconds)), '{:<8}'.format(cfg['name'])])

    def class_row(self):
        means = np.mean(self.accuracy_of_class, axis=0)
        return '[' + ''.join('{:0<5.3}, '.format(a) for a in means) + ']\t{:<8}'.format(self.model_config['name'])


def summary(tallies, repeating):
    """The closing table: mean and standard deviation of the test accuracy over the repetitions, mean
    run time, then the mean per-class accuracies (the content of train.py:289-306)."""
    header = '\t'.join('{:<8}'.format(h) for h in
                       ('DATASET', 'train_size', 'valid_size', 'RESULTS', 'STD', 'TRAIN_TIME', 'NAME'))
    return '\n'.join(['', 'REPEAT\t{}'.format(repeating), header] + [t.row() for t in tallies] +
                     [t.class_row() for t in tallies])


def main(argv=None):
    global args
    from config import load_configuration
    configuration, args = load_configuration(argv)
    tallies = [Tally(cfg) for cfg in configuration['model_list']]
    for repetition in range(configuration['repeating']):
        for tally in tallies:
            cfg = tally.model_config
            # a run is seeded by the model's current seed; the seed of its next repetition is drawn from
            # that same stream first (train.py:269-275)
            seed = cfg['random_seed']
            np.random.seed(seed)
            cfg['random_seed'] = np.random.randint(1, 1073741824 + 1)
            tf.set_random_seed(seed)
            with tf.Session(config={'intra_op_parallelism_threads': cfg['threads']}) as sess:
                accuracy, accuracy_of_class, _, seconds = train(cfg, sess, seed)
            tally.add(accuracy, accuracy_of_class, seconds)
        print('repeated ', repetition, 'rounds')
    print(summary(tallies, configuration['repeating']))


if __name__ == '__main__':
    main()
