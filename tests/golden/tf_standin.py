"""Eager float64 stand-in for the slice of TensorFlow 1.x that the reference's graph code calls.

BUILD-CONTAINER TOOL ONLY.  `make_arith_goldens.py` installs this module as ``tensorflow`` in
``sys.modules`` and then imports the reference's own ``cfl.layers``, ``cfl.ops``, ``cfl.models.{base,
blocks, dist, cfl}`` from ``/root/reference`` and runs their constructors (``Dist(...)``, ``CFL(...)``)
on seeded inputs.  Everything the reference does *between* TF primitives -- which head feeds which side
of a distance, reshape orders, softmax axes, loss composition, pos_weight / regulariser placement,
variable scopes and names, which variables each optimiser owns -- is therefore executed from the
reference's own lines.  The TF primitives themselves are restated here, a few lines each, from
TensorFlow's documented definitions (float64, torch CPU tensors so that `tf.gradients` and
`Optimizer.minimize` come from autograd): this file pins the COMPOSITION, not TensorFlow's kernels.

Semantics restated (TF 1.x docs):
  variable_scope     names nest with '/', `reuse=True` is inherited, `reuse=False` == inherit;
                     get_variable in reuse mode must find the variable, otherwise must not;
                     default_name scopes are uniquified ('Conv', 'Conv_1', ...) and the counters of the
                     sub-scopes are cleared when a scope is left (so a re-entered scope starts at 'Conv')
  get_collection     filters by re.match(scope, item.name)
  regularizer        applied once, when the variable is created; collected in REGULARIZATION_LOSSES
  l2_regularizer(s)  s * sum(w^2) / 2                    xavier_initializer  U(+-sqrt(6/(fan_in+fan_out)))
  l2_normalize       x * rsqrt(max(sum x^2, 1e-12))      moments             population variance
  sigmoid_cross_entropy_with_logits   max(x,0) - x z + log1p(exp(-|x|))
  conv2d 'SAME'      total pad max((ceil(n/s)-1) s + k - n, 0), the extra one at the bottom / right
  conv2d_transpose   the gradient of that conv2d with respect to its input
  maximum / relu     subgradient at ties: maximum -> first argument gets it (x >= y); relu'(0) = 0
  AdamOptimizer      lr_t = lr sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t m / (sqrt(v) + eps); lr, b1, b2, eps
                     carry their float32 values (they are float32 tensors in the reference's graph)
  ExponentialMovingAverage(d)   zero-initialised shadow, shadow -= (1-d)(shadow - value), no debias

A "session run" is modelled by `run_train_ops(ops)`: all gradients are taken first (at the same
pre-update values), then every update is applied -- what one sess.run([op1, op2, ...]) does.
Variable values, Adam slots and EMA shadows persist across `reset_graph()`; everything else (tensors,
collections, scope counters) belongs to one model construction, because eager tensors are values.
"""
import contextlib
import re
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


# ----------------------------------------------------------------------------------------------
# tensors
# ----------------------------------------------------------------------------------------------
class _Dim(int):
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    @property
    def ndims(self):
        return len(self)

    def as_list(self):
        return [int(d) for d in self]

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else _Dim(r)


class _DType(object):
    def __init__(self, name):
        self.name = name

    @property
    def base_dtype(self):
        return self

    def __repr__(self):
        return 'tf.' + self.name


float32 = _DType('float32')
float64 = _DType('float64')
int32 = _DType('int32')
int64 = _DType('int64')
bool_ = _DType('bool')


def _tt(x):
    """torch view of anything tensor-like"""
    if isinstance(x, Tensor):
        return x.t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, (list, tuple)) and any(isinstance(e, Tensor) for e in x):
        return torch.stack([_tt(e) for e in x])
    a = np.asarray(x)
    if a.dtype.kind in 'iu':
        return torch.as_tensor(a.astype(np.int64))
    if a.dtype.kind == 'b':
        return torch.as_tensor(a)
    return torch.as_tensor(a.astype(np.float64))


class Tensor(object):
    """A value.  Arithmetic follows TF's operator overloads (all out-of-place)."""
    __array_priority__ = 1000
    __array_ufunc__ = None

    def __init__(self, t, name=None):
        self.t = t
        self.name = name or 'Tensor'

    # -- shape / dtype surface the reference touches
    def get_shape(self):
        return TensorShape(self.t.shape)

    @property
    def shape(self):
        return TensorShape(self.t.shape)

    @property
    def dtype(self):
        if self.t.dtype == torch.bool:
            return bool_
        return float32 if self.t.dtype.is_floating_point else int32

    def numpy(self):
        return self.t.detach().cpu().numpy()

    # -- operators
    def __add__(self, o): return Tensor(self.t + _tt(o))
    def __radd__(self, o): return Tensor(_tt(o) + self.t)
    def __sub__(self, o): return Tensor(self.t - _tt(o))
    def __rsub__(self, o): return Tensor(_tt(o) - self.t)
    def __mul__(self, o): return Tensor(self.t * _tt(o))
    def __rmul__(self, o): return Tensor(_tt(o) * self.t)
    def __truediv__(self, o): return Tensor(self.t / _tt(o))
    def __rtruediv__(self, o): return Tensor(_tt(o) / self.t)
    def __neg__(self): return Tensor(-self.t)
    def __pow__(self, o): return Tensor(self.t ** _tt(o))

    def __getitem__(self, i):
        return Tensor(self.t[i])

    def __iter__(self):
        return (Tensor(self.t[i]) for i in range(self.t.shape[0]))

    def __len__(self):
        return self.t.shape[0]

    def __repr__(self):
        return '<standin Tensor %s %s>' % (self.name, tuple(self.t.shape))


class Variable(Tensor):
    def __init__(self, t, name, trainable=True):
        Tensor.__init__(self, t, name)
        self.trainable = trainable

    @property
    def op(self):
        return self


def _wrap(t):
    return Tensor(t)


# ----------------------------------------------------------------------------------------------
# graph state
# ----------------------------------------------------------------------------------------------
class GraphKeys(object):
    TRAINABLE_VARIABLES = 'trainable_variables'
    GLOBAL_VARIABLES = 'variables'
    REGULARIZATION_LOSSES = 'regularization_losses'
    SUMMARIES = 'summaries'
    UPDATE_OPS = 'update_ops'


class _State(object):
    def __init__(self):
        self.values = {}          # persistent: variable name -> np.float64 array (current value)
        self.adam = {}            # persistent: (optimizer index, var name) -> (m, v) ; ('pow', idx) -> (b1p, b2p)
        self.ema = {}             # persistent: ema key -> float64 array
        self.initial = {}         # injected initial values (name -> array); otherwise the initializer draws
        self.feeds = {}           # placeholder name -> array
        self.init_rng = np.random.RandomState(0)
        self.rand_rng = np.random.RandomState(0)
        self.reset_graph()

    def reset_graph(self):
        self.vars = {}            # this construction: name -> Variable
        self.collections = {}
        self.scope_stack = [VariableScope('', None)]
        self.name_stack = ['']
        self.scope_counts = {}    # full scope name -> times opened (for default_name uniquification)
        self.n_optimizers = 0
        self.n_ema = 0
        self.draws = []           # (name, array) of every random op, in creation order
        self.created = []         # variable names in creation order


_S = None


def state():
    return _S


def reset_graph():
    """A new model construction: tensors / collections / scope counters go, variable values stay."""
    _S.reset_graph()


def reset_all(seed=0):
    global _S
    _S = _State()
    _S.init_rng = np.random.RandomState(seed)
    _S.rand_rng = np.random.RandomState(seed + 1)


def add_to_collection(key, item):
    _S.collections.setdefault(key, []).append(item)


def get_collection(key, scope=None):
    items = list(_S.collections.get(key, []))
    if scope is None:
        return items
    if isinstance(scope, VariableScope):
        scope = scope.name
    return [it for it in items if hasattr(it, 'name') and re.match(scope, it.name)]


# ----------------------------------------------------------------------------------------------
# variable scopes
# ----------------------------------------------------------------------------------------------
class VariableScope(object):
    def __init__(self, name, reuse):
        self.name = name
        self.reuse = reuse
        self.original_name_scope = name + '/' if name else ''

    def __repr__(self):
        return '<VariableScope %r reuse=%r>' % (self.name, self.reuse)


def get_variable_scope():
    return _S.scope_stack[-1]


_S = _State()


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, reuse=None, **kw):
    cur = _S.scope_stack[-1]
    if name_or_scope is None:
        if reuse is True:
            raise ValueError('reuse=True cannot be used without a name_or_scope')
        if default_name is None:
            raise TypeError('If default_name is None then name_or_scope is required')
        base = (cur.name + '/' if cur.name else '') + default_name
        n = _S.scope_counts.get(base, 0)
        full = base if n == 0 else '%s_%d' % (base, n)
        # TF bumps the counter of the *un-suffixed* name
        _S.scope_counts[base] = n + 1
    elif isinstance(name_or_scope, VariableScope):
        full = name_or_scope.name
        _S.scope_counts[full] = _S.scope_counts.get(full, 0) + 1
    else:
        full = (cur.name + '/' if cur.name else '') + name_or_scope
        _S.scope_counts[full] = _S.scope_counts.get(full, 0) + 1
    if reuse is True:
        r = True
    elif isinstance(name_or_scope, VariableScope) and reuse is None:
        r = name_or_scope.reuse or cur.reuse
    else:
        r = cur.reuse        # None / False inherit
    sc = VariableScope(full, r)
    _S.scope_stack.append(sc)
    _S.name_stack.append(full)
    try:
        yield sc
    finally:
        _S.scope_stack.pop()
        _S.name_stack.pop()
        # leaving a scope clears the default-name counters of its sub-scopes
        pre = full + '/'
        for k in [k for k in _S.scope_counts if k.startswith(pre)]:
            del _S.scope_counts[k]


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    cur = _S.name_stack[-1]
    full = (cur + '/' if cur else '') + (name or default_name or '')
    _S.name_stack.append(full)
    try:
        yield full + '/'
    finally:
        _S.name_stack.pop()


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True,
                 collections=None, **kw):
    sc = _S.scope_stack[-1]
    full = (sc.name + '/' if sc.name else '') + name
    if sc.reuse:
        if full not in _S.vars:
            raise ValueError('Variable %s does not exist, or was not created with tf.get_variable()' % full)
        return _S.vars[full]
    if full in _S.vars:
        raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True?' % full)
    shape = [int(d) for d in (shape if shape is not None else [])]
    if full in _S.values:
        val = np.asarray(_S.values[full], dtype=np.float64)
    elif full in _S.initial:
        val = np.asarray(_S.initial[full], dtype=np.float64)
    else:
        if initializer is None:
            initializer = glorot_uniform_initializer()
        val = np.asarray(initializer(shape), dtype=np.float64)
    assert list(val.shape) == shape, (full, val.shape, shape)
    _S.values[full] = val.copy()
    v = Variable(torch.tensor(val, dtype=DT, requires_grad=True), full + ':0', trainable)
    v.var_name = full
    _S.vars[full] = v
    _S.created.append(full)
    add_to_collection(GraphKeys.GLOBAL_VARIABLES, v)
    if trainable:
        add_to_collection(GraphKeys.TRAINABLE_VARIABLES, v)
    if regularizer is not None:
        loss = regularizer(v)
        if loss is not None:
            loss.name = full + '/Regularizer/l2_regularizer:0'
            add_to_collection(GraphKeys.REGULARIZATION_LOSSES, loss)
    return v


# ----------------------------------------------------------------------------------------------
# initialisers / regularisers
# ----------------------------------------------------------------------------------------------
def _fans(shape):
    if len(shape) < 1:
        return 1.0, 1.0
    if len(shape) == 1:
        return float(shape[0]), float(shape[0])
    if len(shape) == 2:
        return float(shape[0]), float(shape[1])
    rf = 1.0
    for d in shape[:-2]:
        rf *= d
    return float(shape[-2]) * rf, float(shape[-1]) * rf


def xavier_initializer(uniform=True, seed=None, dtype=None):
    def init(shape, dtype=None, partition_info=None):
        fi, fo = _fans(shape)
        lim = np.sqrt(6.0 / (fi + fo))
        return _S.init_rng.uniform(-lim, lim, size=shape)
    return init


glorot_uniform_initializer = xavier_initializer


def zeros_initializer(dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.zeros(shape)


def ones_initializer(dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.ones(shape)


def constant_initializer(value=0.0, dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.full(shape, value, dtype=np.float64)


def l2_regularizer(scale, scope=None):
    def reg(w):
        return Tensor(float(scale) * (_tt(w) ** 2).sum() / 2.0)
    return reg


# ----------------------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------------------
def convert_to_tensor(x, dtype=None, name=None):
    return x if isinstance(x, Tensor) else Tensor(_tt(x))


def constant(value, dtype=None, shape=None, name=None):
    return Tensor(_tt(value).to(DT) if not isinstance(value, Tensor) else value.t)


def placeholder(dtype, shape=None, name=None):
    if name in _S.feeds:
        return Tensor(_tt(_S.feeds[name]), name)
    return None     # an unfed placeholder: anything computed from it must stay unused


def placeholder_with_default(default, shape=None, name=None):
    if name in _S.feeds:
        return Tensor(_tt(_S.feeds[name]), name)
    return default if isinstance(default, Tensor) or default is None else Tensor(_tt(default), name)


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(int(a) for a in axis)
    return int(axis)


def reduce_sum(x, axis=None, keepdims=False, name=None, keep_dims=None, reduction_indices=None):
    axis = reduction_indices if axis is None else axis
    keepdims = keepdims or bool(keep_dims)
    t = _tt(x)
    a = _axes(axis)
    return Tensor(t.sum() if a is None else t.sum(dim=a, keepdim=keepdims))


def reduce_mean(x, axis=None, keepdims=False, name=None, keep_dims=None, reduction_indices=None):
    axis = reduction_indices if axis is None else axis
    keepdims = keepdims or bool(keep_dims)
    t = _tt(x)
    a = _axes(axis)
    return Tensor(t.mean() if a is None else t.mean(dim=a, keepdim=keepdims))


def reshape(x, shape, name=None):
    return Tensor(_tt(x).reshape([int(s) for s in shape]))


def square(x, name=None): return Tensor(_tt(x) ** 2)
def sqrt(x, name=None): return Tensor(torch.sqrt(_tt(x)))
def subtract(x, y, name=None): return Tensor(_tt(x) - _tt(y))
def multiply(x, y, name=None): return Tensor(_tt(x) * _tt(y))
def matmul(a, b, name=None): return Tensor(_tt(a) @ _tt(b))
def add_n(xs, name=None):
    out = _tt(xs[0])
    for x in xs[1:]:
        out = out + _tt(x)
    return Tensor(out)


def maximum(x, y, name=None):
    """tie -> the gradient goes to x (TF's MaximumGrad mask is x >= y)"""
    a, b = _tt(x).to(DT), _tt(y).to(DT)
    a, b = torch.broadcast_tensors(a, b)
    return Tensor(torch.where(a >= b, a, b))


def minimum(x, y, name=None):
    a, b = _tt(x).to(DT), _tt(y).to(DT)
    a, b = torch.broadcast_tensors(a, b)
    return Tensor(torch.where(a <= b, a, b))


def clip_by_value(x, lo, hi, name=None):
    return minimum(maximum(x, lo), hi)


def greater(x, y, name=None): return Tensor(_tt(x) > _tt(y))
def less_equal(x, y, name=None): return Tensor(_tt(x) <= _tt(y))


def cast(x, dtype, name=None):
    t = _tt(x)
    if dtype in (int32, int64):
        return Tensor(t.to(torch.int64))
    return Tensor(t.to(DT))


def ones_like(x, dtype=None, name=None): return Tensor(torch.ones_like(_tt(x), dtype=DT))
def zeros_like(x, dtype=None, name=None): return Tensor(torch.zeros_like(_tt(x), dtype=DT))


def concat(values, axis, name=None):
    if isinstance(values, Tensor):
        return values
    return Tensor(torch.cat([_tt(v) for v in values], dim=int(axis)))


def stack(values, axis=0, name=None):
    if all(not isinstance(v, Tensor) for v in values):
        return list(values)            # a shape made of python ints (array_ops.stack(output_shape))
    return Tensor(torch.stack([_tt(v) for v in values], dim=int(axis)))


def unstack(x, num=None, axis=0, name=None):
    return [Tensor(t) for t in torch.unbind(_tt(x), dim=int(axis))]


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    t = _tt(value)
    if isinstance(num_or_size_splits, int):
        size = t.shape[axis] // num_or_size_splits
        assert size * num_or_size_splits == t.shape[axis]
        parts = torch.split(t, size, dim=axis)
    else:
        parts = torch.split(t, [int(s) for s in num_or_size_splits], dim=axis)
    return [Tensor(p) for p in parts]


def slice_(x, begin, size, name=None):
    t = _tt(x)
    idx = []
    for d, (b, s) in enumerate(zip(begin, size)):
        b, s = int(b), int(s)
        idx.append(slice(b, t.shape[d] if s == -1 else b + s))
    return Tensor(t[tuple(idx)])


def expand_dims(x, axis, name=None): return Tensor(_tt(x).unsqueeze(int(axis)))
def transpose(x, perm=None, name=None):
    t = _tt(x)
    return Tensor(t.permute(*[int(p) for p in perm]) if perm is not None else t.t())


def tile(x, multiples, name=None): return Tensor(_tt(x).repeat(*[int(m) for m in multiples]))
def range_(*a, **k): return Tensor(torch.arange(*[int(x) for x in a]))


def gather_nd(params, indices, name=None):
    p, i = _tt(params), _tt(indices).long()
    return Tensor(p[tuple(i[..., k] for k in range(i.shape[-1]))])


def shape(x, name=None):
    return [int(d) for d in _tt(x).shape]


def group(*ops, **kw):
    return list(ops)


def identity(x, name=None):
    return x


def map_fn(fn, elems, **kw):
    return stack([fn(e) for e in elems], 0)


def _record_draw(name, arr):
    _S.draws.append((name, arr))
    return Tensor(torch.tensor(arr))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    key = name or 'random_normal'
    if key in _S.feeds:
        return _record_draw(key, np.asarray(_S.feeds[key], dtype=np.float64))
    return _record_draw(key, mean + stddev * _S.rand_rng.randn(*[int(s) for s in shape]))


def random_uniform(shape, minval=0, maxval=None, dtype=None, seed=None, name=None):
    key = name or 'random_uniform'
    shape = [int(s) for s in shape]
    if key in _S.feeds:
        return _record_draw(key, np.asarray(_S.feeds[key]))
    if dtype in (int32, int64):
        return _record_draw(key, _S.rand_rng.randint(int(minval), int(maxval), size=shape).astype(np.int64))
    hi = 1.0 if maxval is None else maxval
    return _record_draw(key, _S.rand_rng.uniform(minval, hi, size=shape))


def gradients(ys, xs, grad_ys=None, name=None):
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    total = sum(_tt(y).sum() for y in ys)
    gs = torch.autograd.grad(total, [_tt(x) for x in xs], create_graph=True, allow_unused=True)
    return [None if g is None else Tensor(g) for g in gs]


# ---- nn ------------------------------------------------------------------------------------
def _relu(x, name=None):
    t = _tt(x)
    return Tensor(torch.where(t > 0, t, torch.zeros_like(t)))      # relu'(0) = 0


def _sigmoid(x, name=None): return Tensor(torch.sigmoid(_tt(x)))
def _tanh(x, name=None): return Tensor(torch.tanh(_tt(x)))


def _softmax(logits, axis=-1, name=None, dim=None):
    axis = dim if dim is not None else axis
    return Tensor(torch.softmax(_tt(logits), dim=int(axis)))


def _sce(_sentinel=None, labels=None, logits=None, name=None):
    x, z = _tt(logits), _tt(labels)
    return Tensor(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs())))


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    axis = dim if axis is None else axis
    t = _tt(x)
    ss = (t * t).sum(dim=_axes(axis), keepdim=True)
    return Tensor(t * torch.rsqrt(torch.clamp(ss, min=epsilon)))


def _moments(x, axes, shift=None, name=None, keep_dims=False):
    t = _tt(x)
    a = _axes(axes)
    mean = t.mean(dim=a, keepdim=keep_dims)
    var = ((t - t.mean(dim=a, keepdim=True)) ** 2).mean(dim=a, keepdim=keep_dims)
    return Tensor(mean), Tensor(var)


def _bias_add(value, bias, data_format=None, name=None):
    return Tensor(_tt(value) + _tt(bias))


def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def _conv2d(input, filter, strides, padding, use_cudnn_on_gpu=True, data_format='NHWC', name=None):
    assert padding == 'SAME' and data_format in (None, 'NHWC')
    x, w = _tt(input), _tt(filter)              # NHWC, HWIO
    sh, sw = int(strides[1]), int(strides[2])
    _, pt, pb = _same_pads(x.shape[1], w.shape[0], sh)
    _, pl, pr = _same_pads(x.shape[2], w.shape[1], sw)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), stride=(sh, sw))
    return Tensor(y.permute(0, 2, 3, 1))


def _conv2d_transpose(value, filter, output_shape, strides, padding='SAME', data_format='NHWC', name=None):
    """gradient of conv2d(., filter[kh,kw,out_c,in_c], strides, SAME) with respect to its input, evaluated at
    `value`: the full transposed convolution cropped by the forward op's leading pads."""
    assert padding == 'SAME' and data_format in (None, 'NHWC')
    x, w = _tt(value), _tt(filter)
    sh, sw = int(strides[1]), int(strides[2])
    OH, OW = int(output_shape[1]), int(output_shape[2])
    _, pt, _pb = _same_pads(OH, w.shape[0], sh)
    _, pl, _pr = _same_pads(OW, w.shape[1], sw)
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=(sh, sw))
    need_h, need_w = pt + OH, pl + OW
    if full.shape[2] < need_h or full.shape[3] < need_w:     # rows the forward op pads but never reads
        full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    y = full[:, :, pt:pt + OH, pl:pl + OW]
    return Tensor(y.permute(0, 2, 3, 1))


def _fully_connected(inputs, num_outputs, activation_fn=_relu, normalizer_fn=None, normalizer_params=None,
                     weights_initializer=None, weights_regularizer=None,
                     biases_initializer=zeros_initializer(), biases_regularizer=None, reuse=None,
                     variables_collections=None, outputs_collections=None, trainable=True, scope=None):
    """tf.contrib.layers.fully_connected: variables 'weights' [in, out] and 'biases' under the
    default scope name 'fully_connected'; y = act(x W + b)."""
    with variable_scope(scope, 'fully_connected', [inputs], reuse=reuse):
        n_in = int(inputs.get_shape()[-1])
        W = get_variable('weights', [n_in, num_outputs], initializer=weights_initializer or xavier_initializer(),
                         regularizer=weights_regularizer, trainable=trainable)
        y = matmul(inputs, W)
        if biases_initializer is not None:
            b = get_variable('biases', [num_outputs], initializer=biases_initializer,
                             regularizer=biases_regularizer, trainable=trainable)
            y = _bias_add(y, b)
        if activation_fn is not None:
            y = activation_fn(y)
        return y


def _flatten(inputs, outputs_collections=None, scope=None):
    t = _tt(inputs)
    return Tensor(t.reshape(t.shape[0], -1))


# ---- training ------------------------------------------------------------------------------
class _MinimizeOp(object):
    def __init__(self, opt, loss, var_list):
        self.opt, self.loss, self.var_list = opt, loss, list(var_list)


class AdamOptimizer(object):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name='Adam'):
        # _prepare(): the python floats become tensors and are cast to the variables' dtype -- float32 in the
        # reference.  This float64 evaluation keeps those float32 VALUES (beta2 = fl32(0.999) = 0.99900001287...),
        # so that (1 - beta2) is what the reference's fp32 kernel multiplies g^2 with.
        self.lr, self.b1, self.b2, self.eps = (float(np.float32(x)) for x in (learning_rate, beta1, beta2, epsilon))
        self.index = _S.n_optimizers
        _S.n_optimizers += 1

    def minimize(self, loss, global_step=None, var_list=None, **kw):
        if var_list is None:
            var_list = get_collection(GraphKeys.TRAINABLE_VARIABLES)
        return _MinimizeOp(self, loss, var_list)


def compute_gradients(op):
    """{var name: d loss / d var (float64 array; None where the loss does not reach the variable)}"""
    ts = [v.t for v in op.var_list]
    gs = torch.autograd.grad(_tt(op.loss), ts, retain_graph=True, allow_unused=True)
    return {v.var_name: (None if g is None else g.detach().numpy().copy()) for v, g in zip(op.var_list, gs)}


def run_train_ops(ops):
    """One sess.run([...ops]): every gradient at the pre-update values, then every Adam apply.
    Returns [{var: gradient}] per op."""
    grads = [compute_gradients(op) for op in ops]
    for op, g in zip(ops, grads):
        o = op.opt
        # beta1_power / beta2_power are float32 variables in TF (created from python floats), whatever the
        # dtype of the trained variables
        b1p, b2p = _S.adam.get(('pow', o.index), (np.float32(o.b1), np.float32(o.b2)))
        lr_t = o.lr * np.sqrt(1.0 - np.float64(b2p)) / (1.0 - np.float64(b1p))
        for name, gv in g.items():
            if gv is None:          # apply_gradients skips variables the loss does not reach
                continue
            m, v = _S.adam.get((o.index, name), (np.zeros_like(gv), np.zeros_like(gv)))
            m = o.b1 * m + (1.0 - o.b1) * gv
            v = o.b2 * v + (1.0 - o.b2) * gv * gv
            _S.adam[(o.index, name)] = (m, v)
            _S.values[name] = _S.values[name] - lr_t * m / (np.sqrt(v) + o.eps)
        _S.adam[('pow', o.index)] = (np.float32(b1p * np.float32(o.b1)), np.float32(b2p * np.float32(o.b2)))
    return grads


class ExponentialMovingAverage(object):
    def __init__(self, decay, num_updates=None, zero_debias=False, name='ExponentialMovingAverage'):
        self.decay = decay
        self._avg = {}

    def apply(self, var_list=None):
        keys = []
        for t in var_list:
            key = _S.n_ema
            _S.n_ema += 1
            self._avg[id(t)] = key
            keys.append((key, t))
        return ('ema_apply', self, keys)

    def average(self, t):
        key = self._avg[id(t)]
        return Tensor(torch.tensor(np.asarray(_S.ema.get(key, 0.0), dtype=np.float64)))


def run_ema_ops(ops):
    for _, ema, keys in ops:
        for key, t in keys:
            sh = np.asarray(_S.ema.get(key, 0.0), dtype=np.float64)
            _S.ema[key] = sh - (1.0 - ema.decay) * (sh - t.numpy())


# ---- summaries (inert: they only have to accept their arguments) ------------------------------
class _Summary(object):
    def __init__(self, name):
        self.name = (_S.name_stack[-1] + '/' if _S.name_stack[-1] else '') + str(name)


def _summary_scalar(name, tensor, collections=None, family=None):
    s = _Summary(name)
    add_to_collection(GraphKeys.SUMMARIES, s)
    return s


def _summary_merge(inputs, collections=None, name=None):
    return list(inputs)


# ----------------------------------------------------------------------------------------------
# module assembly
# ----------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def add_arg_scope(fn):
    return fn


class PartitionedVariable(object):
    pass


def install():
    """Register this stand-in as `tensorflow` (and the sub-modules the reference imports from)."""
    me = sys.modules[__name__]
    tf = _mod('tensorflow')
    for k in ('float32', 'float64', 'int32', 'int64', 'GraphKeys', 'variable_scope', 'name_scope', 'get_variable',
              'get_variable_scope', 'get_collection', 'add_to_collection', 'xavier_initializer',
              'zeros_initializer', 'ones_initializer', 'constant_initializer', 'convert_to_tensor', 'constant',
              'placeholder', 'placeholder_with_default', 'reduce_sum', 'reduce_mean', 'reshape', 'square', 'sqrt',
              'subtract', 'multiply', 'matmul', 'add_n', 'maximum', 'minimum', 'clip_by_value', 'greater',
              'less_equal', 'cast', 'ones_like', 'zeros_like', 'concat', 'stack', 'unstack', 'split',
              'expand_dims', 'transpose', 'tile', 'gather_nd', 'shape', 'group', 'identity', 'map_fn',
              'random_normal', 'random_uniform', 'gradients', 'Variable'):
        setattr(tf, k, getattr(me, k))
    tf.bool = bool_
    tf.slice = slice_
    tf.range = range_
    tf.nn = _mod('tensorflow.nn', relu=_relu, sigmoid=_sigmoid, tanh=_tanh, softmax=_softmax,
                 sigmoid_cross_entropy_with_logits=_sce, l2_normalize=_l2_normalize, moments=_moments,
                 bias_add=_bias_add, conv2d=_conv2d, conv2d_transpose=_conv2d_transpose)
    tf.train = _mod('tensorflow.train', AdamOptimizer=AdamOptimizer,
                    ExponentialMovingAverage=ExponentialMovingAverage)
    tf.summary = _mod('tensorflow.summary', scalar=_summary_scalar, histogram=_summary_scalar,
                      image=_summary_scalar, merge=_summary_merge)
    layers = _mod('tensorflow.contrib.layers', xavier_initializer=xavier_initializer, l2_regularizer=l2_regularizer,
                  fully_connected=_fully_connected, flatten=_flatten)
    initializers = _mod('tensorflow.contrib.layers.python.layers.initializers', xavier_initializer=xavier_initializer)
    utils = _mod('tensorflow.contrib.layers.python.layers.utils',
                 get_variable_collections=lambda cs, name: None,
                 collect_named_outputs=lambda collections, alias, outputs: outputs)
    pl = _mod('tensorflow.contrib.layers.python.layers', initializers=initializers, utils=utils)
    lp = _mod('tensorflow.contrib.layers.python', layers=pl)
    layers.python = lp
    fops = _mod('tensorflow.contrib.framework.python.ops', add_arg_scope=add_arg_scope)
    fpy = _mod('tensorflow.contrib.framework.python', ops=fops)
    framework = _mod('tensorflow.contrib.framework', python=fpy, add_arg_scope=add_arg_scope)
    tf.contrib = _mod('tensorflow.contrib', layers=layers, framework=framework)
    ops = _mod('tensorflow.python.framework.ops', convert_to_tensor=convert_to_tensor, get_collection=get_collection,
               add_to_collection=add_to_collection, GraphKeys=GraphKeys)
    pframework = _mod('tensorflow.python.framework', ops=ops)
    array_ops = _mod('tensorflow.python.ops.array_ops', shape=shape, stack=stack)
    init_ops = _mod('tensorflow.python.ops.init_ops', zeros_initializer=zeros_initializer,
                    ones_initializer=ones_initializer)
    vs = _mod('tensorflow.python.ops.variable_scope', variable_scope=variable_scope, get_variable=get_variable)
    variables = _mod('tensorflow.python.ops.variables', PartitionedVariable=PartitionedVariable)
    pops = _mod('tensorflow.python.ops', array_ops=array_ops, init_ops=init_ops, nn=tf.nn, variable_scope=vs,
                variables=variables)
    tf.python = _mod('tensorflow.python', framework=pframework, ops=pops)
    mnist = _mod('tensorflow.examples.tutorials.mnist')
    tut = _mod('tensorflow.examples.tutorials', mnist=mnist)
    tf.examples = _mod('tensorflow.examples', tutorials=tut)
    reg = {
        'tensorflow': tf, 'tensorflow.nn': tf.nn, 'tensorflow.train': tf.train, 'tensorflow.summary': tf.summary,
        'tensorflow.contrib': tf.contrib, 'tensorflow.contrib.layers': layers,
        'tensorflow.contrib.layers.python': lp, 'tensorflow.contrib.layers.python.layers': pl,
        'tensorflow.contrib.layers.python.layers.initializers': initializers,
        'tensorflow.contrib.layers.python.layers.utils': utils,
        'tensorflow.contrib.framework': framework, 'tensorflow.contrib.framework.python': fpy,
        'tensorflow.contrib.framework.python.ops': fops,
        'tensorflow.python': tf.python, 'tensorflow.python.framework': pframework,
        'tensorflow.python.framework.ops': ops, 'tensorflow.python.ops': pops,
        'tensorflow.python.ops.array_ops': array_ops, 'tensorflow.python.ops.init_ops': init_ops,
        'tensorflow.python.ops.nn': tf.nn, 'tensorflow.python.ops.variable_scope': vs,
        'tensorflow.python.ops.variables': variables,
        'tensorflow.examples': tf.examples, 'tensorflow.examples.tutorials': tut,
        'tensorflow.examples.tutorials.mnist': mnist,
    }
    sys.modules.update(reg)
    # scipy.misc.{imread,imsave,imresize} no longer exist; cfl/utils.py and cfl/input_data.py import the names
    import scipy
    misc = types.ModuleType('scipy.misc')
    misc.imread = misc.imsave = misc.imresize = None
    sys.modules['scipy.misc'] = misc
    scipy.misc = misc
    return tf
