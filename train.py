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


def train(model_config, sess, seed, data_split=None):
    verbose = bool(args is not None and args.verbose)
    very_begining = time.time()
    print('',
          'name           : {}'.format(model_config['name']),
          'dataset        : {}'.format(model_config['dataset']),
          'train_size     : {}'.format(model_config['train_size']),
          'learning_rate  : {}'.format(model_config['learning_rate']),
          'feature        : {}'.format(model_config['feature']),
          sep='\n')
    if model_config['Model'] != 'GCN':
        raise ValueError("model_config['Model'] must be 'GCN' in this build, but is {} now".format(
            model_config['Model']))

    if data_split:
        adj, features = data_split['adj'], data_split['features']
        y_train, y_val, y_test = data_split['y_train'], data_split['y_val'], data_split['y_test']
        train_mask, val_mask, test_mask = data_split['train_mask'], data_split['val_mask'], data_split['test_mask']
    else:
        adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = load_data(
            model_config['dataset'], train_size=model_config['train_size'],
            validation_size=model_config['validation_size'], model_config=model_config,
            shuffle=model_config['shuffle'])

    if not sparse.issparse(features):
        raise ValueError('this build feeds bag-of-words features as a sparse triple')
    features = sparse_to_tuple(features)
    support = build_support(adj, model_config['conv'])
    num_supports = len(support)

    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32, name='support' + str(i)) for i in range(num_supports)],
        'features': tf.sparse_placeholder(tf.float32, name='features'),
        'labels': tf.placeholder(tf.int32, name='labels', shape=(None, y_train.shape[1])),
        'labels_mask': tf.placeholder(tf.int32, name='labels_mask'),
        'dropout': tf.placeholder_with_default(0., name='dropout', shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32, name='num_features_nonzero'),
    }
    model = GCN_MLP(model_config, placeholders, input_dim=features[2][1])
    sess.run(tf.global_variables_initializer())

    train_feed_dict = construct_feed_dict(features, support, y_train, train_mask, placeholders)
    train_feed_dict.update({placeholders['dropout']: model_config['dropout']})
    valid_feed_dict = construct_feed_dict(features, support, y_val, val_mask, placeholders)
    test_feed_dict = construct_feed_dict(features, support, y_test, test_mask, placeholders)

    max_valid_acc = 0
    max_train_acc = 0
    t_test = time.time()
    test_cost, test_acc, test_acc_of_class, prediction = sess.run(
        [model.loss, model.accuracy, model.accuracy_of_class, model.prediction], feed_dict=test_feed_dict)
    test_duration = time.time() - t_test
    timer = 0
    step_times = []

    if model_config['train']:
        print('training...')
        for step in range(model_config['epochs']):
            _sync()
            t = time.time()
            sess.run(model.opt_op, feed_dict=train_feed_dict)
            _sync()
            t = time.time() - t
            timer += t
            step_times.append(t)
            train_loss, train_acc, train_summary = sess.run([model.loss, model.accuracy, model.summary],
                                                            feed_dict=train_feed_dict)
            improved = False
            if model_config['validate']:
                valid_loss, valid_acc, valid_summary = sess.run([model.loss, model.accuracy, model.summary],
                                                                feed_dict=valid_feed_dict)
                if valid_acc > max_valid_acc:
                    max_valid_acc = valid_acc
                    improved = True
            elif train_acc > max_train_acc:
                max_train_acc = train_acc
                improved = True
            if improved:
                t_test = time.time()
                test_cost, test_acc, test_acc_of_class = sess.run(
                    [model.loss, model.accuracy, model.accuracy_of_class], feed_dict=test_feed_dict)
                test_duration = time.time() - t_test
                prediction = sess.run(model.prediction, train_feed_dict)
                if verbose:
                    print('*', end='')
            if verbose:
                print("Epoch: {:04d}".format(step), "train_loss= {:.3f}".format(train_loss),
                      "train_acc= {:.3f}".format(train_acc), end=' ')
                if model_config['validate']:
                    print("val_loss=", "{:.3f}".format(valid_loss), "val_acc= {:.3f}".format(valid_acc), end=' ')
                print("time=", "{:.5f}".format(t))
        else:
            print("Optimization Finished!")
        print("Test set results:", "cost=", "{:.5f}".format(test_cost), "accuracy=", "{:.5f}".format(test_acc),
              "time=", "{:.5f}".format(test_duration))
        print("accuracy of each class=", test_acc_of_class)
        if step_times:
            print("epoch ms (optimiser step): median {:.3f}  min {:.3f}".format(
                1e3 * float(np.median(step_times)), 1e3 * float(np.min(step_times))))
    print("Total time={}s".format(time.time() - very_begining))
    return test_acc, test_acc_of_class, prediction, time.time() - very_begining


def main(argv=None):
    global args
    from config import load_configuration
    configuration, args = load_configuration(argv)
    model_list = configuration['model_list']
    acc = [[] for _ in model_list]
    acc_of_class = [[] for _ in model_list]
    duration = [[] for _ in model_list]
    for r in range(configuration['repeating']):
        for i, model_config in enumerate(model_list):
            seed = model_config['random_seed']
            np.random.seed(seed)
            model_config['random_seed'] = np.random.randint(1, 1073741824 + 1)
            tf.set_random_seed(seed)
            with tf.Session(config={'intra_op_parallelism_threads': model_config['threads']}) as sess:
                test_acc, test_acc_of_class, prediction, t = train(model_config, sess, seed)
            acc[i].append(test_acc)
            acc_of_class[i].append(test_acc_of_class)
            duration[i].append(t)
        print('repeated ', r, 'rounds')

    acc_means = np.mean(acc, axis=1)
    acc_stds = np.std(acc, axis=1)
    acc_of_class_means = np.mean(acc_of_class, axis=1)
    duration = np.mean(duration, axis=1)
    print()
    print("REPEAT\t{}".format(configuration['repeating']))
    print("{:<8}\t{:<8}\t{:<8}\t{:<8}\t{:<8}\t{:<8}\t{:<8}".format(
        'DATASET', 'train_size', 'valid_size', 'RESULTS', 'STD', 'TRAIN_TIME', 'NAME'))
    for model_config, acc_mean, acc_std, t in zip(model_list, acc_means, acc_stds, duration):
        print("{:<8}\t{:<8}\t{:<8}\t{:<8.6f}\t{:<8.6f}\t{:<8.2f}\t{:<8}".format(
            model_config['dataset'], str(model_config['train_size']) + ' per class',
            str(model_config['validation_size']), acc_mean, acc_std, t, model_config['name']))
    for model_config, acc_of_class_mean in zip(model_list, acc_of_class_means):
        print('[', end='')
        for a in acc_of_class_mean:
            print('{:0<5.3}'.format(a), end=', ')
        print(']', end='')
        print('\t{:<8}'.format(model_config['name']))


if __name__ == '__main__':
    main()
