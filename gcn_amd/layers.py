"""Layers of the reference (gcn/layers.py) on the HIP operators.

``sparse_dropout``, ``dot`` and ``GraphConvolution`` keep the reference's names, argument
names and defaults.  Every product runs in ``libgcn_hip.so``; torch supplies memory, streams
and the autograd tape.
"""
import torch

from . import graph, inits, sparse

_LAYER_UIDS = {}
_stream_counter = [0]


def get_layer_uid(layer_name=''):
    """Unique layer IDs (gcn/layers.py:9-16)."""
    _LAYER_UIDS[layer_name] = _LAYER_UIDS.get(layer_name, 0) + 1
    return _LAYER_UIDS[layer_name]


def next_stream_id():
    """A fresh random-stream id for one dropout draw."""
    _stream_counter[0] += 1
    return _stream_counter[0]


def relu(x):
    """tf.nn.relu; GraphConvolution fuses it into the product's epilogue when it is the activation."""
    return torch.relu(x)


# --------------------------------------------------------------------------- #
# autograd bindings of the HIP operators
# --------------------------------------------------------------------------- #
class _SpmmFn(torch.autograd.Function):
    """out = act(A . B); backward: dB = A^T . (dOut gated by the ReLU).

    With ``out_dropout`` (a sparse.OutDropout: the NEXT layer's input dropout) the product stores
    dropout(act(A . B)) and the consumer -- _GatedGemmFn -- applies the backward of both the dropout and
    the ReLU in the store of its own gradient product: the gradient that arrives here is already the
    one of A . B."""

    @staticmethod
    def forward(ctx, a, b, fuse_relu, out_dropout=None):
        out = sparse.spmm(a, b, relu=fuse_relu, out_dropout=out_dropout)
        ctx.a = a
        ctx.gate = fuse_relu and out_dropout is None
        if ctx.gate:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        if ctx.gate:
            out, = ctx.saved_tensors
            grad = sparse.dropout(grad, 1.0, 0, 0, gate=out)      # grad where out > 0, else 0
        # the support keeps its default plan (every row of up to 256 entries summed in storage order)
        return None, sparse.spmm(ctx.a.transpose(), grad.contiguous()), None, None


class _SpmmDropoutFn(torch.autograd.Function):
    """out = sparse_dropout(X) . W with the mask regenerated (not stored) in the backward:
    dW = sparse_dropout(X)^T . dOut through the transposed copy of X and its entry ids."""

    @staticmethod
    def forward(ctx, x, w, keep_prob, seed, stream_id):
        ctx.x, ctx.keep_prob, ctx.seed, ctx.stream_id = x, keep_prob, seed, stream_id
        return sparse.spmm_dropout(x, w, keep_prob, seed, stream_id)

    @staticmethod
    def backward(ctx, grad):
        gw = sparse.spmm_dropout(ctx.x.transpose(for_gradient=True), grad, ctx.keep_prob, ctx.seed, ctx.stream_id)
        return None, gw, None, None, None


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep_prob, seed, stream_id):
        ctx.keep_prob, ctx.seed, ctx.stream_id = keep_prob, seed, stream_id
        return sparse.dropout(x, keep_prob, seed, stream_id)

    @staticmethod
    def backward(ctx, grad):
        return sparse.dropout(grad, ctx.keep_prob, ctx.seed, ctx.stream_id), None, None, None


class _GemmFn(torch.autograd.Function):
    """out = A . W on the MFMA kernel; dA = dOut . W^T on it too; dW = A^T . dOut (a [k x n]
    reduction over all nodes) on the transposed-left kernel."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return sparse.gemm(a, w)

    @staticmethod
    def backward(ctx, grad):
        a, w = ctx.saved_tensors
        ga = sparse.gemm(grad, w, trans_b=True) if ctx.needs_input_grad[0] else None
        gw = sparse.gemm_tn(a, grad) if ctx.needs_input_grad[1] else None
        return ga, gw


class _DroppedInputGemmFn(torch.autograd.Function):
    """dropout(x) . W for a layer input x that needs no gradient (the features), the dropout applied to x as
    the product loads it: tf.nn.dropout (gcn/layers.py:185) and tf.matmul (gcn/layers.py:198-199) in one
    kernel.  The decisions live in a bit matrix (1/32 of the bytes of x); the weight gradient
    dropout(x)^T . dOut reads x and the same bits, so the dropped matrix is never written."""

    @staticmethod
    def forward(ctx, x, w, bits, keep_prob):
        ctx.save_for_backward(x, bits)
        ctx.keep_prob = keep_prob
        return sparse.gemm_masked(x, bits, keep_prob, w)

    @staticmethod
    def backward(ctx, grad):
        x, bits = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise RuntimeError('the dropped layer input of _DroppedInputGemmFn takes no gradient')
        d_w = sparse.gemm_tn_masked(x, bits, ctx.keep_prob, grad.contiguous()) if ctx.needs_input_grad[1] else None
        return None, d_w, None, None


class _GatedGemmFn(torch.autograd.Function):
    """h_dropped . W where h_dropped = dropout(relu(z)) came out of a product's store (_SpmmFn with
    ``out_dropout``).  The gradient it returns for its left operand is the gradient of z:
    (dP . W^T) / keep where h_dropped > 0, else 0 -- the dropout's and the ReLU's backward in the store
    of the product dP . W^T (h_dropped is positive exactly where the unit was active and kept) instead of
    two passes over the activations."""

    @staticmethod
    def forward(ctx, h_dropped, w, keep_prob):
        ctx.save_for_backward(h_dropped, w)
        ctx.keep_prob = keep_prob
        return sparse.gemm(h_dropped, w)

    @staticmethod
    def backward(ctx, grad):
        h, w = ctx.saved_tensors
        grad = grad.contiguous()
        d_h = sparse.gemm_gated(grad, w, h, ctx.keep_prob, trans_b=True) if ctx.needs_input_grad[0] else None
        d_w = sparse.gemm_tn(h, grad) if ctx.needs_input_grad[1] else None
        return d_h, d_w, None


class DroppedSparse(object):
    """A sparse matrix with a pending ``sparse_dropout``: the mask is applied inside the next product."""

    def __init__(self, csr, keep_prob, seed, stream_id):
        self.csr, self.keep_prob, self.seed, self.stream_id = csr, keep_prob, seed, stream_id


# --------------------------------------------------------------------------- #
# the reference's free functions
# --------------------------------------------------------------------------- #
def sparse_dropout(x, keep_prob, noise_shape, seed=None, stream_id=None):
    """Dropout for sparse tensors (gcn/layers.py:19-25): entry k of ``x`` is kept iff
    floor(keep_prob + u_k) == 1 and kept entries are scaled by 1/keep_prob.  ``noise_shape``
    must be the number of stored entries, as in the reference."""
    keep_prob = float(keep_prob)
    if noise_shape is not None:
        n = int(noise_shape[0]) if hasattr(noise_shape, '__len__') else int(noise_shape)
        if n != x.nnz:
            raise ValueError('noise_shape %d does not match the %d stored entries' % (n, x.nnz))
    if keep_prob >= 1.0:
        return x
    if not keep_prob > 0.0:
        raise ValueError('keep_prob must be in (0, 1]')
    return DroppedSparse(x, keep_prob, graph.get_random_seed() if seed is None else seed,
                         next_stream_id() if stream_id is None else stream_id)


def dense_dropout(x, keep_prob, seed=None, stream_id=None):
    """tf.nn.dropout(x, keep_prob) as called at gcn/layers.py:185."""
    keep_prob = float(keep_prob)
    if keep_prob >= 1.0:
        return x
    return _DropoutFn.apply(x, keep_prob, graph.get_random_seed() if seed is None else seed,
                            next_stream_id() if stream_id is None else stream_id)


def dot(x, y, sparse=False, fuse_relu=False, out_dropout=None, dropped_input=None):
    """Wrapper for the two matmuls of the reference (gcn/layers.py:28-34): ``sparse=True`` is
    tf.sparse_tensor_dense_matmul (x a DeviceCSR, possibly with a pending dropout), otherwise
    tf.matmul.  This build's extensions: ``fuse_relu`` -- ReLU in the product's epilogue;
    ``out_dropout`` (sparse product with fuse_relu) -- the next layer's input dropout in its store;
    ``dropped_input`` (dense product) -- the keep probability of such a dropout already carried by x."""
    if sparse:
        if isinstance(x, DroppedSparse):
            res = _SpmmDropoutFn.apply(x.csr, y, x.keep_prob, x.seed, x.stream_id)
            return torch.relu(res) if fuse_relu else res
        return _SpmmFn.apply(x, y, fuse_relu, out_dropout if fuse_relu else None)
    res = _GatedGemmFn.apply(x, y, dropped_input) if dropped_input is not None else _GemmFn.apply(x, y)
    return torch.relu(res) if fuse_relu else res


# --------------------------------------------------------------------------- #
# layer classes
# --------------------------------------------------------------------------- #
def _resolve(node):
    """The value behind a layer input: a bound placeholder, an upstream layer's last output or a tensor."""
    if isinstance(node, graph.Placeholder):
        if getattr(node, 'value', None) is None:
            raise ValueError('placeholder %r was not fed' % (node.name,))
        return node.value
    if isinstance(node, Layer):
        return node.output
    return node


class Layer(object):
    """Base layer (gcn/layers.py:37-86): name, vars, input width discovery, ``__call__`` -> ``_call``."""

    def __init__(self, input, **kwargs):
        allowed_kwargs = {'name', 'logging', 'use_theta'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        name = kwargs.get('name')
        if not name:
            name = self.__class__.__name__.lower()
        self.name = name
        self.vars = {}
        self.logging = kwargs.get('logging', False)
        self.sparse_inputs = False
        self.input = input
        self.output = None
        if isinstance(input, graph.Placeholder) and input.sparse:
            self.input_dim = input._my_input_dim          # the hack of gcn/layers.py:66-67
        elif isinstance(input, graph.Placeholder):
            self.input_dim = input.shape[1]
        elif isinstance(input, Layer):
            self.input_dim = input.output_dim
        elif isinstance(input, sparse.DeviceCSR):
            self.input_dim = input.n_cols
        else:
            self.input_dim = int(input.shape[1])

    def _call(self, x):
        return x

    def __call__(self, x=None):
        outputs = self._call(_resolve(self.input) if x is None else x)
        self.output = outputs
        return outputs


class GraphConvolution(Layer):
    """Graph convolution layer (gcn/layers.py:139-214): dropout -> X.W -> A_hat.(XW) -> (+bias) -> act."""

    def __init__(self, input, output_dim, placeholders, dropout=0., sparse_inputs=False, act=relu, bias=False,
                 featureless=False, use_theta=False, **kwargs):
        super(GraphConvolution, self).__init__(input, **kwargs)
        if use_theta:
            raise ValueError('use_theta (conv == "chebytheta") is outside the GCN propagation path of this build')
        self.dropout = placeholders['dropout'] if dropout else 0.
        self.act = act
        self.support = placeholders['support']
        self.sparse_inputs = sparse_inputs
        self.featureless = featureless
        self.bias = bias
        self.use_theta = use_theta
        self.output_dim = output_dim
        self.num_features_nonzero = placeholders['num_features_nonzero']    # helper for sparse dropout
        for i in range(len(self.support)):
            self.vars['weights_' + str(i)] = inits.glorot([self.input_dim, output_dim],
                                                         name=self.name + '/weights_' + str(i))
        if self.bias:
            self.vars['bias'] = inits.zeros([output_dim], name=self.name + '/bias')
        self.stream_id = None          # set per run by the model: which dropout stream this layer draws
        # set per run by the model when two layers follow each other: this layer's product stores its
        # output with the next layer's input dropout applied (a sparse.OutDropout), and the next layer is
        # told the keep probability its input already carries
        self.next_dropout = None
        self.input_dropped = None

    def _dropout_rate(self):
        d = self.dropout
        if isinstance(d, graph.Placeholder):
            d = d.default if getattr(d, 'value', None) is None else d.value
        return float(d)

    def can_drop_output(self):
        """Whether this layer's activation and the next layer's dropout can go into its product's store."""
        return (self.act is relu) and not self.bias and len(self.support) == 1

    def takes_dropped_input(self):
        return not self.sparse_inputs and not self.featureless

    def _call(self, x):
        keep_prob = 1.0 - self._dropout_rate()
        stream_id = self.stream_id if self.stream_id is not None else next_stream_id()
        dropped_input = self.input_dropped if self.takes_dropped_input() else None
        if dropped_input is not None:
            pass                                           # the producer stored x already dropped
        elif self.sparse_inputs:
            nnz = _resolve(self.num_features_nonzero) if isinstance(self.num_features_nonzero, graph.Placeholder) \
                else self.num_features_nonzero
            x = sparse_dropout(x, keep_prob, nnz, stream_id=stream_id)
        else:
            x = dense_dropout(x, keep_prob, stream_id=stream_id)

        fuse = (self.act is relu) and not self.bias and len(self.support) == 1
        supports = []
        for i in range(len(self.support)):
            if not self.featureless:
                pre_sup = dot(x, self.vars['weights_' + str(i)], sparse=self.sparse_inputs,
                              dropped_input=dropped_input)
            else:
                pre_sup = self.vars['weights_' + str(i)]
            supports.append(dot(_resolve(self.support[i]), pre_sup, sparse=True, fuse_relu=fuse,
                                out_dropout=self.next_dropout if fuse else None))
        output = supports[0]
        for s in supports[1:]:
            output = output + s
        if self.bias:
            output = output + self.vars['bias']
        return output if fuse else self.act(output)
