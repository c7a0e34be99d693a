"""The model of the reference (gcn/models.py:6-128) on the HIP layers.

``GCN_MLP(model_config, placeholders, input_dim)`` builds the layer stack named by
``model_config['connection']`` and exposes ``outputs``, ``prediction``, ``loss``, ``accuracy``,
``accuracy_of_class``, ``opt_op`` and ``summary`` as fetch handles for ``Session.run``.
Only graph-convolution layers ('c') are in scope; the other connection letters raise.
"""
from copy import copy

import numpy as np
import torch

from . import graph, sparse
from .layers import GraphConvolution, relu
from .metrics import masked_accuracy


class TfAdam(object):
    """tf.train.AdamOptimizer(learning_rate) as built at gcn/models.py:35-38 (beta1 0.9,
    beta2 0.999, epsilon 1e-8).  TensorFlow folds the bias corrections into the step size and
    adds epsilon to sqrt(v) uncorrected: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);
    var -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, params, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step_size(self, t):
        return self.lr * float(np.sqrt(1.0 - self.b2 ** t)) / (1.0 - self.b1 ** t)

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        lr_t = self.step_size(self.t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.add_(g - m, alpha=1.0 - self.b1)
            v.add_(g * g - v, alpha=1.0 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)

    @torch.no_grad()
    def step_fused(self, grads, lr_t, weight_decay=0.0):
        """One kernel per variable (gcn_adam_f32); ``grads`` exclude the weight-decay term, which the
        kernel adds as weight_decay * var.  ``lr_t`` is a float, or a 0-d device tensor holding
        step_size(t) when the launches are recorded in a HIP graph (the caller keeps ``t``)."""
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            sparse.adam_step(p, g, m, v, lr_t, self.b1, self.b2, self.eps, weight_decay)


class GCN_MLP(object):
    def __init__(self, model_config, placeholders, input_dim):
        self.model_config = model_config
        self.name = model_config['name']
        if not self.name:
            self.name = self.__class__.__name__.lower()
        self.logging = False

        self.vars = {}
        self.layers = []
        self.activations = []
        self.act = [relu for _ in range(len(self.model_config['connection']))]
        self.act[-1] = lambda x: x

        self.placeholders = placeholders
        self.inputs = placeholders['features']
        self.inputs._my_input_dim = input_dim
        self.input_dim = input_dim
        self.output_dim = placeholders['labels'].get_shape().as_list()[1]
        self.global_step = 0
        self.optimizer = None
        self._run_counter = 0
        self._device_cache = {}
        self._new_id = None
        # train steps recorded as HIP graphs (one per distinct feed), replayed with a single launch:
        # on the citation graphs a step is ~30 kernels of a few microseconds each, bound by launch rate
        self.use_hip_graph = bool(model_config.get('hip_graph', False))
        self._recorded = {}

        self.outputs = graph.Fetch(self, 'outputs')
        self.prediction = graph.Fetch(self, 'prediction')
        self.loss = graph.Fetch(self, 'loss')
        self.accuracy = graph.Fetch(self, 'accuracy')
        self.accuracy_of_class = [graph.Fetch(self, 'accuracy_of_class', i) for i in range(self.output_dim)]
        self.opt_op = graph.Fetch(self, 'opt_op')
        self.summary = graph.Fetch(self, 'summary')

        self.build()

    # ------------------------------------------------------------------ #
    def build(self):
        """Layer construction of gcn/models.py:43-67."""
        classes = {'c': GraphConvolution}
        layer_type = []
        for ch in self.model_config['connection']:
            if ch not in classes:
                raise ValueError('connection letter "%s": only graph convolutions ("c") are part of this build' % ch)
            layer_type.append(classes[ch])
        layer_size = copy(self.model_config['layer_size'])
        layer_size.insert(0, self.input_dim)
        layer_size.append(self.output_dim)
        is_sparse = bool(getattr(self.placeholders['features'], 'sparse', False))
        prev = self.inputs
        for li, (output_dim, layer_cls, act) in enumerate(zip(layer_size[1:], layer_type, self.act)):
            layer = layer_cls(input=prev, output_dim=output_dim, placeholders=self.placeholders, act=act,
                              dropout=True, sparse_inputs=is_sparse, logging=self.logging,
                              use_theta=self.model_config['conv'] == 'chebytheta',
                              name='%s/graphconvolution_%d' % (self.name, li))
            self.layers.append(layer)
            is_sparse = False
            prev = layer
        # every variable is a view into one buffer, so that the weight-decay term of the loss is one pass
        # over it inside the kernel that also averages the loss and accuracy rows
        total = sum(var.numel() for layer in self.layers for var in layer.vars.values())
        first = next(var for layer in self.layers for var in layer.vars.values())
        self._flat_weights = torch.empty(total, dtype=torch.float32, device=first.device)
        pos = 0
        for layer in self.layers:
            for key, var in list(layer.vars.items()):
                view = self._flat_weights[pos:pos + var.numel()].view(var.shape)
                with torch.no_grad():
                    view.copy_(var)
                layer.vars[key] = view.requires_grad_(True)
                pos += var.numel()
                self.vars[layer.name + '/' + key] = layer.vars[key]
        self.optimizer = TfAdam(self._trainable(), self.model_config['learning_rate'])

    def _trainable(self):
        return [var for layer in self.layers for var in layer.vars.values()]

    # ------------------------------------------------------------------ #
    def _bind(self, feed_dict):
        """Put the fed values behind the placeholders, uploading each host object once.

        The device copy of a fed host array or triple is cached by object identity: feeding the same
        object again (as train.py does every epoch) costs nothing, and a value CHANGED IN PLACE after its
        first feed is not seen -- feed a new array (``mask.copy()``) instead.  The reference path of this
        build never mutates a fed value; its label-expansion code (gcn/utils.py:554) does."""
        for ph in list(self.placeholders.values()):
            for p in (ph if isinstance(ph, (list, tuple)) else [ph]):
                if isinstance(p, graph.Placeholder):
                    p.value = None
        for ph, val in feed_dict.items():
            if not isinstance(ph, graph.Placeholder):
                raise TypeError('feed_dict keys must be placeholders')
            ph.value = self._to_device(ph, val)
        self._bind_vertex_order(feed_dict)

    def _bind_vertex_order(self, feed_dict):
        """A support made with ``preprocess_adj(reorder=...)`` is the matrix of the relabelled graph: the
        features, labels and mask fed beside it (in the dataset's order) are permuted to that order once per
        fed object, and ``_evaluate`` puts the outputs back (``self._new_id``; None without a reorder)."""
        self._new_id = None
        orders = [(getattr(v, 'new_id', None), getattr(v, 'old_id', None)) for v in feed_dict.values()]
        orders = [o for o in orders if o[0] is not None]
        if not orders:
            return
        supports = self.placeholders['support']
        if len(orders) != len(supports) or any(o[0] is not orders[0][0] and not torch.equal(o[0], orders[0][0])
                                               for o in orders):
            raise ValueError('every support of a model must carry the same vertex order')
        new_id, old_id = orders[0]
        self._new_id = new_id.to(torch.int64)
        old = old_id.to(torch.int64)
        for name in ('features', 'labels', 'labels_mask'):
            ph = self.placeholders[name]
            val = ph.value
            if val is None:
                continue
            hit = self._device_cache.get(('reordered', id(val)))
            if hit is None or hit[0] is not val:
                if isinstance(val, sparse.DeviceCSR):
                    moved = sparse.permute_rows(val, old_id)
                    moved.perm = None        # sparse dropout draws by entry position in the permuted matrix
                else:
                    moved = val[old].contiguous()
                hit = (val, moved)
                self._device_cache[('reordered', id(val))] = hit
            ph.value = hit[1]

    def _dataset_order(self, rows):
        """Rows of a per-node result in the dataset's order (undoes the support's vertex order)."""
        return rows if self._new_id is None else rows[self._new_id]

    def _to_device(self, ph, val):
        if ph.sparse:
            if isinstance(val, sparse.DeviceCSR):
                return val
            csr = getattr(val, 'csr', None)
            if csr is not None:
                return csr
            key = id(val)
            hit = self._device_cache.get(key)
            if hit is None or hit[0] is not val:
                hit = (val, sparse.triple_to_device_csr(val))
                self._device_cache[key] = hit
            return hit[1]
        if ph.name in ('dropout', 'num_features_nonzero'):
            return val
        if isinstance(val, torch.Tensor):
            return val.cuda()
        key = id(val)
        hit = self._device_cache.get(key)
        if hit is None or hit[0] is not val:
            arr = np.asarray(val)
            dtype = torch.int32 if ph.dtype == graph.int32 else torch.float32
            hit = (val, torch.as_tensor(arr.astype(np.int32 if dtype == torch.int32 else np.float32)).cuda())
            self._device_cache[key] = hit
        return hit[1]

    def _forward(self, stream_base=None):
        """gcn/models.py:54-69: every layer's dropout draws its own stream of this run (``stream_base``:
        0 when the run's share of the stream id is added on the device, as in a recorded step)."""
        self.activations = [self.inputs.value]
        hidden = self.inputs.value
        base = self._run_counter * 64 if stream_base is None else stream_base
        for li, layer in enumerate(self.layers):
            layer.stream_id = base + li
        self._plan_fused_dropout()
        for layer in self.layers:
            hidden = layer(hidden)
            self.activations.append(hidden)
        return hidden

    def _plan_fused_dropout(self):
        """Where a ReLU layer feeds a dense-input layer that drops its input, the dropout (the same
        decisions: stream of the consuming layer, element id row * width + column) is applied in the
        store of the producing layer's product, and its backward in the store of the consumer's gradient
        product."""
        for layer in self.layers:
            layer.next_dropout = layer.input_dropped = None
        for producer, consumer in zip(self.layers, self.layers[1:]):
            keep = 1.0 - consumer._dropout_rate()
            if keep < 1.0 and producer.can_drop_output() and consumer.takes_dropped_input():
                producer.next_dropout = sparse.OutDropout(keep, graph.get_random_seed(), consumer.stream_id)
                consumer.input_dropped = keep

    def _mask_mean(self, mask):
        """mean(mask) as a host float, computed once per fed mask."""
        hit = self._device_cache.get(('mask_mean', id(mask)))
        if hit is None or hit[0] is not mask:
            hit = (mask, float(mask.to(torch.float32).mean().item()))
            self._device_cache[('mask_mean', id(mask))] = hit
        return hit[1]

    def _loss_and_accuracy(self, stats):
        """[loss, accuracy] on the device: gcn/models.py:92-99 (weight_decay * l2_loss(var) over every
        variable of every layer plus the masked cross entropy) and 106-108, from the loss kernel's rows."""
        wd = self.model_config['weight_decay']
        flat = self._flat_weights
        end = flat.data_ptr() + 4 * flat.numel()
        if all(flat.data_ptr() <= v.data_ptr() < end for v in self._trainable()):
            return sparse.loss_summary(stats, flat, wd)
        # a variable was replaced by a tensor of the caller's: sum over the variables as they are
        means = stats.mean(dim=0)
        reg = sum(wd * (var * var).sum() / 2 for var in self._trainable())
        return torch.stack([means[0] + reg, means[1]])

    # ------------------------------------------------------------------ #
    # the train step as one HIP graph
    # ------------------------------------------------------------------ #
    _RECORDABLE = {'opt_op', 'loss', 'accuracy', 'outputs', 'prediction', 'summary'}

    def _feed_signature(self, keys, feed_dict):
        sig = [tuple(keys)]
        for ph, val in sorted(feed_dict.items(), key=lambda kv: str(kv[0].name)):
            bound = ph.value
            sig.append((ph.name, id(bound) if isinstance(bound, (torch.Tensor, sparse.DeviceCSR)) else repr(bound)))
        return tuple(sig)

    def _record_train_step(self, keys):
        """Capture forward + loss + backward + Adam for the currently bound feed.  Everything that
        allocates with a synchronisation (plans, transposed copies) is touched by one eager step first;
        the model state that step changes is put back before the capture."""
        labels = self.placeholders['labels'].value
        mask = self.placeholders['labels_mask'].value
        mask_mean = self._mask_mean(mask)
        params = self._trainable()
        snapshot = [p.detach().clone() for p in params + self.optimizer.m + self.optimizer.v]
        dev = params[0].device
        rec = {'offset': torch.zeros(1, dtype=torch.int64, device=dev),
               'lr_t': torch.zeros((), dtype=torch.float32, device=dev)}

        def body():
            with sparse.stream_offset(rec['offset']):          # 64 * run counter, added to the stream ids by the kernels
                with torch.enable_grad():
                    outputs = self._forward(stream_base=0)
                    stats, dlogits = sparse.softmax_xent(outputs.detach(), labels, mask, mask_mean)
                    grads = torch.autograd.grad(outputs, params, grad_outputs=dlogits)
                with torch.no_grad():
                    rec['outputs'] = outputs.detach()
                    if 'loss' in keys or 'accuracy' in keys:
                        rec['summary2'] = self._loss_and_accuracy(stats)     # with the weights of this forward
                    if 'prediction' in keys:
                        rec['prediction'] = torch.softmax(outputs.detach(), dim=1)
                self.optimizer.step_fused(grads, rec['lr_t'], self.model_config['weight_decay'])

        import warnings
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), warnings.catch_warnings():
            # the weights were created on the default stream and the recording runs on its own: expected
            warnings.filterwarnings('ignore', message='.*AccumulateGrad.*')
            body()                                          # eager: builds plans, transposes, workspaces
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for p, saved in zip(params + self.optimizer.m + self.optimizer.v, snapshot):
                p.copy_(saved)
        rec['graph'] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(rec['graph']), warnings.catch_warnings():
            warnings.filterwarnings('ignore', message='.*AccumulateGrad.*')
            body()
        return rec

    def _evaluate_recorded(self, keys, feed_dict):
        sig = self._feed_signature(keys, feed_dict)
        rec = self._recorded.get(sig)
        if rec is None:
            rec = self._record_train_step(keys)
            self._recorded[sig] = rec
        self._run_counter += 1
        self.optimizer.t += 1
        rec['offset'].fill_(64 * self._run_counter)
        rec['lr_t'].fill_(self.optimizer.step_size(self.optimizer.t))
        rec['graph'].replay()
        self.global_step += 1
        out = {'opt_op': None, 'summary': None}
        if 'loss' in keys or 'accuracy' in keys:
            loss, accuracy = rec['summary2'].tolist()                    # one read-back for both
            if 'loss' in keys:
                out['loss'] = np.float32(loss)
            if 'accuracy' in keys:
                out['accuracy'] = np.float32(accuracy)
        if 'outputs' in keys:
            out['outputs'] = graph.to_numpy(self._dataset_order(rec['outputs']))
        if 'prediction' in keys:
            out['prediction'] = graph.to_numpy(self._dataset_order(rec['prediction']))
        return out

    def _evaluate(self, keys, feed_dict):
        self._bind(feed_dict)
        if self.use_hip_graph and 'opt_op' in keys and set(keys) <= self._RECORDABLE:
            return self._evaluate_recorded(keys, feed_dict)
        self._run_counter += 1
        labels = self.placeholders['labels'].value
        mask = self.placeholders['labels_mask'].value
        train = 'opt_op' in keys
        out = {}
        need_stats = train or 'loss' in keys or 'accuracy' in keys
        with torch.set_grad_enabled(train):
            outputs = self._forward()
            stats = loss = None
            if need_stats:      # loss rows, accuracy rows and d loss / d logits in one pass over the logits
                stats, dlogits = sparse.softmax_xent(outputs.detach(), labels, mask, self._mask_mean(mask),
                                                     want_grad=train)
                if 'loss' in keys or 'accuracy' in keys:
                    with torch.no_grad():
                        summary = self._loss_and_accuracy(stats)       # with the weights of this run's forward
            if train:
                params = self._trainable()
                grads = torch.autograd.grad(outputs, params, grad_outputs=dlogits)
                self.optimizer.t += 1
                self.optimizer.step_fused(grads, self.optimizer.step_size(self.optimizer.t),
                                          self.model_config['weight_decay'])
                self.global_step += 1
                out['opt_op'] = None
        with torch.no_grad():
            if 'outputs' in keys:
                out['outputs'] = graph.to_numpy(self._dataset_order(outputs))
            if 'prediction' in keys:
                out['prediction'] = graph.to_numpy(self._dataset_order(torch.softmax(outputs, dim=1)))   # models.py:89-90
            if 'loss' in keys or 'accuracy' in keys:
                loss, accuracy = summary.tolist()                                               # one read-back for both
                if 'loss' in keys:
                    out['loss'] = np.float32(loss)
                if 'accuracy' in keys:
                    out['accuracy'] = np.float32(accuracy)                                      # models.py:106-108
            if 'accuracy_of_class' in keys:
                m = mask.to(torch.float32)
                out['accuracy_of_class'] = [
                    np.float32(masked_accuracy(outputs, labels, m * labels[:, i].to(torch.float32)).item())
                    for i in range(self.output_dim)]                                            # models.py:110-113
            if 'summary' in keys:
                out['summary'] = None
        return out

    # ------------------------------------------------------------------ #
    def save(self, sess=None):
        """Checkpointing of the reference (models.py:115-128) is never called by its driver; kept as a
        plain state-dict dump for completeness of the interface."""
        if not sess:
            raise AttributeError('Session not provided.')
        path = 'tmp/%s.pt' % self.name
        torch.save({k: v.detach().cpu() for k, v in self.vars.items()}, path)
        print('Model saved in file: %s' % path)

    def load(self, sess=None):
        if not sess:
            raise AttributeError('Session not provided.')
        path = 'tmp/%s.pt' % self.name
        state = torch.load(path, weights_only=True)
        with torch.no_grad():
            for k, v in state.items():
                self.vars[k].copy_(v)
        print('Model restored from file: %s' % path)
