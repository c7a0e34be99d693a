"""Feed/fetch plumbing that stands where TensorFlow 1.x's graph API stands in the reference.

The reference builds a static graph from placeholders (train.py:144-154) and evaluates
tensors with ``sess.run(fetches, feed_dict)`` (train.py:192-234).  Here a placeholder is a
named slot, a fetch is a named result of a model, and ``Session.run`` binds the feed,
runs the model eagerly on the GPU and returns host values -- same call shapes, no graph
compiler.  Device copies of fed sparse triples are cached per object, so feeding the
same ``SparseTensorValue`` every step (as train.py does) uploads it once.
"""
import collections

import numpy as np

SparseTensorValue = collections.namedtuple('SparseTensorValue', ['indices', 'values', 'dense_shape'])

float32 = 'float32'
int32 = 'int32'

_graph_seed = [0]


def set_random_seed(seed):
    """tf.set_random_seed(seed) of train.py:275: seeds weight init and dropout streams."""
    _graph_seed[0] = int(seed)


def get_random_seed():
    return _graph_seed[0]


class Placeholder(object):
    def __init__(self, dtype, shape=None, name=None, sparse=False, default=None):
        self.dtype, self.shape, self.name, self.sparse, self.default = dtype, shape, name, sparse, default
        self._my_input_dim = None           # the reference hangs the sparse input width here (models.py:22)

    def get_shape(self):
        return _Shape(self.shape)

    def __repr__(self):
        return '<Placeholder %s%s>' % (self.name, ' sparse' if self.sparse else '')


class _Shape(object):
    def __init__(self, shape):
        self._shape = shape

    def as_list(self):
        return list(self._shape) if self._shape is not None else None

    def __getitem__(self, i):
        return self._shape[i]


def sparse_placeholder(dtype, name=None):
    return Placeholder(dtype, None, name, sparse=True)


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def placeholder_with_default(default, shape=(), name=None):
    return Placeholder(float32, shape, name, default=default)


class Fetch(object):
    """Handle of one model output (``model.loss``, ``model.opt_op`` ...)."""

    def __init__(self, model, key, index=None):
        self.model, self.key, self.index = model, key, index

    def __repr__(self):
        return '<Fetch %s%s>' % (self.key, '' if self.index is None else '[%d]' % self.index)


class Session(object):
    """``sess.run`` of the reference's driver.  ``config`` (thread counts) is accepted and ignored."""

    def __init__(self, config=None):
        self.config = config

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flat = []

        def walk(f):
            if isinstance(f, (list, tuple)):
                return [walk(g) for g in f]
            if f is None:
                return None
            if not isinstance(f, Fetch):
                raise TypeError('cannot fetch %r' % (f,))
            flat.append(f)
            return f

        tree = walk([fetches] if single else list(fetches))
        models = []
        for f in flat:
            if f.model not in models:
                models.append(f.model)
        results = {}
        for m in models:
            keys = sorted({f.key for f in flat if f.model is m})
            results[id(m)] = m._evaluate(keys, feed_dict or {})

        def build(node):
            if isinstance(node, list):
                return [build(n) for n in node]
            if node is None:
                return None
            v = results[id(node.model)][node.key]
            return v if node.index is None else v[node.index]

        out = build(tree)
        return out[0] if single else out


def global_variables_initializer():
    """Variables are initialised when a model is built; fetching this is a no-op."""
    return None


def to_numpy(x):
    import torch
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)
