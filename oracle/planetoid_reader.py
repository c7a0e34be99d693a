"""Data-only reader for the Planetoid ``ind.<dataset>.<name>`` files.

TEST INFRASTRUCTURE ONLY.  The files are Python-2 pickles.  This module never
hands them to ``pickle``: it walks the opcode stream with
``pickletools.genops`` (a pure tokenizer) and rebuilds the few value shapes
the files contain -- ints, strings, tuples, lists, dicts, numpy arrays, scipy
CSR matrices, ``defaultdict(list)`` -- from literal data.  No name found in a
file is ever imported or called; an opcode or global outside the short
allow-list below raises ``ValueError``.

The files are the ones the reference's loader opens at
``gcn/utils.py:113-122`` (x, y, tx, ty, allx, ally, graph).
"""
import pickletools

import numpy as np
import scipy.sparse as sp

_MARK = object()


class _Global(object):
    def __init__(self, name):
        self.name = name


class _NdArray(object):
    """Placeholder created by ``numpy.core.multiarray._reconstruct``."""
    value = None


class _Dtype(object):
    def __init__(self, code):
        self.code = code
        self.order = '|'

    def numpy(self):
        order = self.order if self.order in '<>' else ''
        return np.dtype(order + self.code)


class _Csr(object):
    """Placeholder created by NEWOBJ on ``scipy.sparse.csr.csr_matrix``."""
    value = None


_ALLOWED_GLOBALS = {
    'numpy.core.multiarray _reconstruct',
    'numpy ndarray',
    'numpy dtype',
    'scipy.sparse.csr csr_matrix',
    'collections defaultdict',
    '__builtin__ list',
}


def _resolve(v):
    if isinstance(v, _NdArray) or isinstance(v, _Csr):
        if v.value is None:
            raise ValueError('object used before its state was set')
        return v.value
    return v


def _reduce(func, args):
    if not isinstance(func, _Global):
        raise ValueError('REDUCE on a non-global')
    if func.name == 'numpy.core.multiarray _reconstruct':
        return _NdArray()
    if func.name == 'numpy dtype':
        code = args[0]
        if not isinstance(code, str) or len(code) > 4:
            raise ValueError('unexpected dtype code %r' % (code,))
        return _Dtype(code)
    if func.name == 'collections defaultdict':
        if len(args) != 1 or not isinstance(args[0], _Global) or args[0].name != '__builtin__ list':
            raise ValueError('only defaultdict(list) is accepted')
        return {}
    raise ValueError('REDUCE on %s is not accepted' % func.name)


def _build(obj, state):
    if isinstance(obj, _Dtype):
        # (version, byteorder, subdescr, names, fields, itemsize, alignment, flags)
        obj.order = state[1]
        return obj
    if isinstance(obj, _NdArray):
        _version, shape, dtype, fortran, raw = state
        buf = raw.encode('latin1') if isinstance(raw, str) else bytes(raw)
        arr = np.frombuffer(buf, dtype=dtype.numpy()).copy()
        arr = arr.reshape(shape, order='F' if fortran else 'C')
        obj.value = arr
        return obj
    if isinstance(obj, _Csr):
        st = {k: _resolve(v) for k, v in state.items()}
        shape = tuple(int(s) for s in st['_shape'])
        obj.value = sp.csr_matrix((st['data'], st['indices'], st['indptr']), shape=shape)
        return obj
    raise ValueError('BUILD on an unexpected object')


def read_planetoid_file(path):
    """Return the value stored in one ``ind.*`` file as plain data."""
    with open(path, 'rb') as f:
        blob = f.read()
    stack = []
    memo = {}

    def pop_to_mark():
        items = []
        while True:
            v = stack.pop()
            if v is _MARK:
                break
            items.append(v)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(blob):
        n = op.name
        if n == 'PROTO':
            continue
        elif n == 'STOP':
            break
        elif n == 'GLOBAL':
            if arg not in _ALLOWED_GLOBALS:
                raise ValueError('global %r is not accepted' % (arg,))
            stack.append(_Global(arg))
        elif n in ('BINPUT', 'LONG_BINPUT'):
            memo[arg] = stack[-1]
        elif n in ('BINGET', 'LONG_BINGET'):
            stack.append(memo[arg])
        elif n == 'MARK':
            stack.append(_MARK)
        elif n in ('BININT', 'BININT1', 'BININT2', 'SHORT_BINSTRING', 'BINSTRING'):
            stack.append(arg)
        elif n == 'NONE':
            stack.append(None)
        elif n == 'NEWTRUE':
            stack.append(True)
        elif n == 'NEWFALSE':
            stack.append(False)
        elif n == 'EMPTY_TUPLE':
            stack.append(())
        elif n == 'EMPTY_LIST':
            stack.append([])
        elif n == 'EMPTY_DICT':
            stack.append({})
        elif n == 'TUPLE1':
            stack[-1:] = [tuple(stack[-1:])]
        elif n == 'TUPLE2':
            stack[-2:] = [tuple(stack[-2:])]
        elif n == 'TUPLE3':
            stack[-3:] = [tuple(stack[-3:])]
        elif n == 'TUPLE':
            stack.append(tuple(pop_to_mark()))
        elif n == 'APPEND':
            v = stack.pop()
            stack[-1].append(v)
        elif n == 'APPENDS':
            items = pop_to_mark()
            stack[-1].extend(items)
        elif n == 'SETITEMS':
            items = pop_to_mark()
            d = stack[-1]
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif n == 'NEWOBJ':
            args = stack.pop()
            cls = stack.pop()
            if not isinstance(cls, _Global) or cls.name != 'scipy.sparse.csr csr_matrix' or args != ():
                raise ValueError('NEWOBJ is only accepted for csr_matrix()')
            stack.append(_Csr())
        elif n == 'REDUCE':
            args = stack.pop()
            func = stack.pop()
            stack.append(_reduce(func, args))
        elif n == 'BUILD':
            state = stack.pop()
            stack[-1] = _build(stack[-1], state)
        else:
            raise ValueError('opcode %s is not accepted' % n)
    if len(stack) != 1:
        raise ValueError('malformed stream')
    return _resolve(stack[0])


def read_index_file(path):
    """One integer per line (``gcn/utils.py:34-39``)."""
    with open(path) as f:
        return [int(line.strip()) for line in f]
