"""TEST INFRASTRUCTURE -- an eager stand-in for the slice of the TensorFlow-1.x API that the reference's
mac_cell.py / ops.py / model.py / config.py touch, so that those files can be EXECUTED UNMODIFIED here
(TensorFlow itself is not installable in this environment).

It is not TensorFlow and makes no attempt to be: every `tf.*` call computes its documented TF-1.x
result immediately on torch CPU tensors (fp64 by default).  What it has to get right for the parity
anchor is small and written down in the TF-1.x API docs:

  tf.variable_scope / tf.get_variable   name stacking ("a/b/weight"), `reuse` inheritance (True is sticky
      downwards, None/False inherit), "already exists" / "does not exist" errors, `default_name`
      uniquification with the per-scope counters reset when the parent scope closes
      (variable_scope.py: _VariableScopeStore.open_variable_scope / close_variable_subscopes)
  initialisers   xavier_initializer = uniform(+-sqrt(6 / (fan_in + fan_out))), zeros, ones, constant,
      random_normal(stddev 1)
  tf.nn.dropout(x, keep)   x / keep * floor(keep + U[0,1));  python-number keep == 1 returns x itself
  tf.nn.softmax   exp(x - max) / sum;  tf.nn.elu   x > 0 ? x : expm1(x);  tf.sequence_mask, tf.tile, ...
  tf.contrib.layers.batch_norm   training: batch moments over all but the last axis, eps 1e-3

Randomness is injectable so the restatement under test can be fed the SAME masks: every uniform draw goes
through `shim.uniform(shape)` and is logged in `shim.state.draws` in call order.

Only tests/ may import this package (tests/test_reference_exec.py, tests/golden/make_reference_golden.py).
"""
import builtins
import math
import sys
import types
from contextlib import contextmanager

import torch

# ------------------------------------------------------------------------------------------------
# state
# ------------------------------------------------------------------------------------------------
float32 = "float32"
float64 = "float64"
int32 = "int32"
int64 = "int64"
bool_ = "bool"


class _State:
    def __init__(self):
        self.reset()

    def reset(self, dtype=torch.float64, seed=0, preset=None, require_grad=False):
        self.dtype = dtype
        self.gen = torch.Generator().manual_seed(seed)
        self.variables = {}          # full name -> tensor, in creation order
        self.preset = dict(preset or {})   # values a "checkpoint restore" would put into the variables
        self.require_grad = require_grad
        self.scope = []              # [(name, reuse)]
        self.counts = {}             # full scope name -> times opened (default_name uniquification)
        self.draws = []              # [("uniform", tensor)] in call order
        self.uniform_hook = None     # shape -> tensor, replaces the generator when set
        self.initial = {}            # full name -> the value the variable was initialised with (EMA shadows start there)
        self.untrainable = {}        # name -> tf.Variable(..., trainable=False)
        self.ema = {}                # variable name -> ExponentialMovingAverage shadow


state = _State()


def shim_reset(**kw):
    state.reset(**kw)


def _dt(dtype):
    if dtype in (None, float32, float64) or dtype is float:
        return state.dtype
    if dtype in (int32,):
        return torch.int32
    if dtype in (int64,) or dtype is int:
        return torch.int64
    if dtype in (bool_,) or dtype is bool:
        return torch.bool
    if isinstance(dtype, torch.dtype):
        return dtype
    raise TypeError("dtype %r" % (dtype,))


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return _w(x)
    return _w(torch.as_tensor(x, dtype=dtype if dtype is not None else (state.dtype if isinstance(x, float) else None)))


def _ints(shape):
    if isinstance(shape, torch.Tensor):
        return [int(v) for v in shape.reshape(-1).tolist()]
    if isinstance(shape, (int,)):
        return [shape]
    return [int(v) for v in shape]


class TFTensor(torch.Tensor):
    """TF tensors are immutable values: `x += y` REBINDS x (mac_cell.py:338 `newMemory += info` must not change
    `memory`; ops.py:1056 `randomTensor += ...` broadcasts a scalar up).  torch's in-place dunders are replaced by
    the out-of-place ones; every tensor the shim creates or is handed is of this class and torch keeps the
    subclass through its ops."""

    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o

    # TF converts the other operand of a tensor operator with ops.convert_to_tensor, which rejects None with
    # ValueError("None values not supported.") -- Python / torch would raise TypeError
    def _none_check(o):
        if o is None:
            raise ValueError("None values not supported.")

    def __mul__(self, o):
        TFTensor._none_check(o)
        return super().__mul__(o)

    def __rmul__(self, o):
        TFTensor._none_check(o)
        return super().__rmul__(o)

    def __add__(self, o):
        TFTensor._none_check(o)
        return super().__add__(o)

    def __radd__(self, o):
        TFTensor._none_check(o)
        return super().__radd__(o)

    def __sub__(self, o):
        TFTensor._none_check(o)
        return super().__sub__(o)

    def get_shape(self):          # `inp.get_shape()[-1]` (ops.py:164)
        return tuple(int(s) for s in self.shape)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # TF's static shape inference rejects incompatible shapes with ValueError while the graph is built
        try:
            return super().__torch_function__(func, types, args, kwargs or {})
        except RuntimeError as e:
            msg = str(e)
            if any(t in msg for t in ("must match the size", "cannot be multiplied", "Sizes of tensors must match",
                                      "is invalid for input of size", "must match except in dimension")):
                # (tf.reshape to a fully specified shape with another element count is a graph-build ValueError too)
                raise ValueError("Dimensions must be equal (shape inference): " + msg) from None
            raise


def _w(t):
    return t if isinstance(t, TFTensor) else t.as_subclass(TFTensor)


def wrap(t):
    """Hand a torch tensor to the reference as a TF value."""
    return _w(t)


def uniform(shape):
    shape = _ints(shape)
    if state.uniform_hook is not None:
        u = torch.as_tensor(state.uniform_hook(tuple(shape)), dtype=state.dtype)
    else:
        u = torch.rand(shape, generator=state.gen, dtype=torch.float64).to(state.dtype)
    state.draws.append(("uniform", u))
    return _w(u)


# ------------------------------------------------------------------------------------------------
# variables and scopes
# ------------------------------------------------------------------------------------------------
def _cur_name():
    return "/".join(n for n, _ in state.scope if n)


def _cur_reuse():
    return any(r is True for _, r in state.scope)


def _unique(prefix):
    cur = _cur_name()
    name = (cur + "/" + prefix) if cur else prefix
    if state.counts.get(name, 0) == 0:
        return prefix
    idx = 1
    while state.counts.get(name + "_%d" % idx, 0) > 0:
        idx += 1
    return prefix + "_%d" % idx


@contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None, **_ignored):
    if name_or_scope is None:
        if default_name is None:
            raise ValueError("variable_scope: name_or_scope and default_name cannot both be None")
        name = _unique(default_name)
    elif hasattr(name_or_scope, "name") and not isinstance(name_or_scope, str):
        # a captured scope object (tf.get_variable_scope()): re-entered, not nested -- only the ROOT scope is ever passed here
        if name_or_scope.name:
            raise NotImplementedError("re-entering a non-root captured scope")
        name = ""
    else:
        name = str(name_or_scope)
    state.scope.append((name, reuse))
    full = _cur_name()
    state.counts[full] = state.counts.get(full, 0) + 1
    try:
        yield full
    finally:
        state.scope.pop()
        pre = full + "/"
        for k in list(state.counts):
            if k.startswith(pre):
                state.counts[k] = 0


def get_variable_scope():
    return types.SimpleNamespace(name=_cur_name(), reuse=_cur_reuse())


class _Init:
    def __init__(self, kind, value=None, stddev=1.0):
        self.kind, self.value, self.stddev = kind, value, stddev

    def __call__(self, shape):
        shape = tuple(_ints(shape))
        if self.kind == "zeros":
            return torch.zeros(shape, dtype=state.dtype)
        if self.kind == "ones":
            return torch.ones(shape, dtype=state.dtype)
        if self.kind == "constant":
            return torch.full(shape, float(self.value), dtype=state.dtype)
        if self.kind == "normal":
            return (torch.randn(shape, generator=state.gen, dtype=torch.float64) * self.stddev).to(state.dtype)
        if self.kind == "xavier":
            # tf.contrib.layers.xavier_initializer(uniform=True) -> variance_scaling(1.0, FAN_AVG, uniform):
            # limit = sqrt(3 * 1 / ((fan_in + fan_out) / 2)); fans per init_ops._compute_fans
            if len(shape) < 1:
                fan_in = fan_out = 1
            elif len(shape) == 1:
                fan_in = fan_out = shape[0]
            elif len(shape) == 2:
                fan_in, fan_out = shape
            else:
                rf = 1
                for s in shape[:-2]:
                    rf *= s
                fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return ((torch.rand(shape, generator=state.gen, dtype=torch.float64) * 2 - 1) * lim).to(state.dtype)
        raise ValueError(self.kind)


def zeros_initializer():
    return _Init("zeros")


def ones_initializer():
    return _Init("ones")


def constant_initializer(value=0.0):
    return _Init("constant", value=value)


def random_normal_initializer(mean=0.0, stddev=1.0):
    return _Init("normal", stddev=stddev)


def get_variable(name, shape=None, initializer=None, dtype=None, trainable=True, **_ignored):
    full = (_cur_name() + "/" + name) if _cur_name() else name
    if _cur_reuse():
        if full not in state.variables:
            raise ValueError("Variable %s does not exist, or was not created with tf.get_variable(). "
                             "Did you mean to set reuse=None in VarScope?" % full)
        return state.variables[full]
    if full in state.variables:
        raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True in VarScope?" % full)
    if isinstance(shape, int):
        shape = (shape,)
    if shape is None and isinstance(initializer, torch.Tensor):
        shape = tuple(initializer.shape)          # tf.get_variable(name, initializer=<tensor>): the tensor gives the shape
    shape = tuple(_ints(shape if shape is not None else ()))
    if full in state.preset:
        v = torch.as_tensor(state.preset[full]).to(state.dtype).clone()
        if tuple(v.shape) != shape:
            raise ValueError("restore: variable %s has shape %s in the checkpoint, the graph wants %s"
                             % (full, tuple(v.shape), shape))
    else:
        if initializer is None:
            initializer = _Init("xavier")      # tf.get_variable's default is glorot_uniform
        v = initializer(shape) if callable(initializer) else torch.as_tensor(initializer, dtype=state.dtype).reshape(shape)
    v = _w(v.detach())
    if state.require_grad and v.is_floating_point():
        v.requires_grad_(True)
    state.variables[full] = v
    state.initial[full] = v.detach().clone()
    return v


class Variable(object):
    """tf.Variable(initial_value, dtype=, trainable=False, name=): the step counter of model.py:617 (eager: a python box)"""

    def __init__(self, initial_value, dtype=None, trainable=True, name=None, **_ignored):
        if trainable:
            raise NotImplementedError("the reference only creates its globalStep this way (model.py:617)")
        self.name = ((_cur_name() + "/") if _cur_name() else "") + (name or "Variable") + ":0"
        self.value = initial_value
        state.untrainable[self.name] = self


def trainable_variables():
    return [types.SimpleNamespace(name=k + ":0", value=v) for k, v in state.variables.items()]


# ------------------------------------------------------------------------------------------------
# array / math ops
# ------------------------------------------------------------------------------------------------
def constant(value, dtype=None, shape=None):
    t = _w(torch.as_tensor(value, dtype=_dt(dtype) if dtype is not None or isinstance(value, float) else None))
    return t.reshape(_ints(shape)) if shape is not None else t


def zeros(shape, dtype=float32):
    return _w(torch.zeros(_ints(shape), dtype=_dt(dtype)))


def ones(shape, dtype=float32):
    return _w(torch.ones(_ints(shape), dtype=_dt(dtype)))


def zeros_like(x):
    return _w(torch.zeros_like(x))


def fill(dims, value):
    v = value.item() if isinstance(value, torch.Tensor) else value
    return _w(torch.full(_ints(dims), v, dtype=torch.int64 if isinstance(v, int) else state.dtype))


def shape(x):
    return _w(torch.tensor(list(x.shape), dtype=torch.int64))


def reshape(x, new_shape):
    return _w(x).reshape(_ints(new_shape))


def concat(values, axis):
    if any(v is None for v in values):
        raise ValueError("None values not supported.")          # ops.convert_to_tensor(None)
    return torch.cat([_t(v) for v in values], dim=axis)


def stack(values, axis=0):
    return torch.stack(list(values), dim=axis)


def unstack(x, num=None, axis=0):
    return list(torch.unbind(x, dim=axis))


def split(value, num_or_size_splits, axis=0):
    if isinstance(num_or_size_splits, int):
        return list(torch.chunk(value, num_or_size_splits, dim=axis))
    return list(torch.split(value, _ints(num_or_size_splits), dim=axis))


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


def tile(x, multiples):
    return x.repeat(*_ints(multiples))


def transpose(x, perm=None):
    return x.permute(*perm) if perm is not None else x.t()


def identity(x, name=None):
    return x


def stop_gradient(x):
    return x.detach()


def cast(x, dtype):
    return _t(x).to(_dt(dtype))


def to_float(x):
    return _t(x).to(state.dtype) if not isinstance(x, float) else _w(torch.tensor(x, dtype=state.dtype))


def to_int32(x):
    return _t(x).to(torch.int32)


def matmul(a, b):
    return torch.matmul(_w(a), b)


def reduce_sum(x, axis=None, keep_dims=False, keepdims=False):
    kd = keep_dims or keepdims
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=kd)


def reduce_mean(x, axis=None, keep_dims=False, keepdims=False):
    kd = keep_dims or keepdims
    x = x if x.is_floating_point() else x.to(state.dtype)
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=kd)


def reduce_max(x, axis=None, keep_dims=False, keepdims=False):
    kd = keep_dims or keepdims
    return x.max() if axis is None else x.max(dim=axis, keepdim=kd).values


def argmax(x, axis=None):
    return torch.argmax(x, dim=axis)


def equal(a, b):
    return _t(a).to(torch.int64) == _t(b).to(torch.int64) if not _t(a).is_floating_point() else _t(a) == _t(b)


def logical_and(a, b):
    return a & b


def maximum(a, b):
    return torch.maximum(_t(a), _t(b))


def minimum(a, b):
    return torch.minimum(_t(a), _t(b))


def floor(x):
    return torch.floor(x)


def div(a, b):
    return a / b


def squared_difference(a, b):
    return (a - b) ** 2


def tanh(x, name=None):
    return torch.tanh(x)


def sigmoid(x, name=None):
    return torch.sigmoid(x)


def exp(x):
    return torch.exp(x)


def log(x):
    return torch.log(x)


def sqrt(x):
    return torch.sqrt(x)


def pow(x, y):  # noqa: A001
    return torch.pow(_t(x), _t(y))


def sin(x):
    return torch.sin(x)


def cos(x):
    return torch.cos(x)


def linspace(start, stop, num):
    return _w(torch.linspace(float(start), float(stop), int(num), dtype=state.dtype))


def range(*a):  # noqa: A001
    return _w(torch.arange(*[int(v) for v in a]))


def meshgrid(x, y):
    return list(torch.meshgrid(x, y, indexing="xy"))


def random_uniform(shape, minval=0, maxval=1, dtype=float32, seed=None):
    return minval + (maxval - minval) * uniform(shape)


def sequence_mask(lengths, maxlen=None):
    lengths = _t(lengths).to(torch.int64)
    maxlen = int(maxlen) if maxlen is not None else int(lengths.max())
    return _w(torch.arange(maxlen).unsqueeze(0) < lengths.unsqueeze(-1))


def cond(pred, true_fn, false_fn):
    return true_fn() if bool(pred) else false_fn()


def global_norm(t_list):
    """clip_ops.global_norm: sqrt(sum_t ||t||_2^2) over the tensors that are not None"""
    return _w(torch.sqrt(sum((t.double() ** 2).sum() for t in t_list if t is not None)))


def clip_by_global_norm(t_list, clip_norm, use_norm=None):
    """clip_ops.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)"""
    norm = global_norm(t_list) if use_norm is None else use_norm
    scale = clip_norm / torch.maximum(norm.double(), torch.tensor(float(clip_norm), dtype=torch.float64))
    return [None if t is None else t * scale.to(t.dtype) for t in t_list], norm


class GraphKeys(object):
    UPDATE_OPS = "update_ops"


def get_collection(key):
    return []          # no tf.layers batch-norm update ops are ever registered by the reference's graph (it uses contrib's with updates_collections=None)


@contextmanager
def control_dependencies(ops_):
    yield              # eager: every op has already run when its python call returns, in program order


def group(*ops_, **_ignored):
    return None


# ------------------------------------------------------------------------------------------------
# tf.train: what MACnet.addOptimizerOp / computeGradients / addTrainingOp (model.py:615-669) call.  Eager: an "op" runs when
# it is built, so ONE pass through those three methods is one training step; optimizer slots live in the optimizer object
# (created once, model.py:618) and EMA shadows in `state.ema` (a graph would create them once and update them per run).
# ------------------------------------------------------------------------------------------------
train = types.ModuleType("tensorflow.train")


class _AdamOptimizer(object):
    """tf.train.AdamOptimizer (training/adam.py + kernels/training_ops.cc ApplyAdam, TF 1.x):
         lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power)        (the powers start at beta1, beta2)
         m <- beta1 m + (1 - beta1) g ;  v <- beta2 v + (1 - beta2) g^2 ;  var <- var - lr_t m / (sqrt(v) + epsilon)
       and, after every variable, beta1_power *= beta1, beta2_power *= beta2 (_finish)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **_ignored):
        self.lr, self.beta1, self.beta2, self.eps = learning_rate, beta1, beta2, epsilon
        self.beta1_power, self.beta2_power = beta1, beta2
        self.m, self.v = {}, {}

    def compute_gradients(self, loss, var_list=None, **_ignored):
        vs = list(var_list) if var_list is not None else trainable_variables()
        gs = torch.autograd.grad(loss, [v.value for v in vs], allow_unused=True, retain_graph=True)
        return [(None if g is None else _w(g.detach()), v) for g, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        lr = float(self.lr)
        lr_t = lr * math.sqrt(1.0 - self.beta2_power) / (1.0 - self.beta1_power)
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                var = v.value
                g = g.detach().to(var.dtype)
                m = self.m.setdefault(v.name, torch.zeros_like(var))
                vv = self.v.setdefault(v.name, torch.zeros_like(var))
                m.mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
                vv.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
                var.data.sub_(lr_t * m / (vv.sqrt() + self.eps))
        self.beta1_power *= self.beta1
        self.beta2_power *= self.beta2
        if global_step is not None:
            global_step.value += 1
        return "train_op"


class _ExponentialMovingAverage(object):
    """tf.train.ExponentialMovingAverage(decay).apply(vars): shadow <- shadow - (1 - decay) (shadow - var); a shadow starts at
    its variable's INITIAL value (moving_averages.py: initialized_value())."""

    def __init__(self, decay, num_updates=None, **_ignored):
        if num_updates is not None:
            raise NotImplementedError("model.py:658 passes the decay only")
        self.decay = decay
        self._vars = []

    def apply(self, var_list=None):
        self._vars = list(var_list) if var_list is not None else trainable_variables()
        with torch.no_grad():
            for v in self._vars:
                key = v.name[:-2] if v.name.endswith(":0") else v.name
                sh = state.ema.get(key)
                if sh is None:
                    sh = state.ema[key] = state.initial[key].clone()
                sh.sub_((1.0 - self.decay) * (sh - v.value.detach()))
        return "ema_op"

    def variables_to_restore(self, moving_avg_variables=None):
        return {(v.name[:-2] if v.name.endswith(":0") else v.name) + "/ExponentialMovingAverage": v for v in self._vars}


train.AdamOptimizer = _AdamOptimizer
train.ExponentialMovingAverage = _ExponentialMovingAverage


# ------------------------------------------------------------------------------------------------
# tf.nn
# ------------------------------------------------------------------------------------------------
nn = types.ModuleType("tensorflow.nn")


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    # nn_ops.dropout (TF 1.x): a python number 1 short-circuits; otherwise
    #   random_tensor = keep_prob + random_uniform(shape);  ret = div(x, keep_prob) * floor(random_tensor)
    if isinstance(keep_prob, (int, float)) and keep_prob == 1:
        return x
    kp = keep_prob if isinstance(keep_prob, torch.Tensor) else torch.tensor(float(keep_prob), dtype=state.dtype)
    random_tensor = kp + uniform(noise_shape if noise_shape is not None else x.shape)
    return (_w(x) / kp) * torch.floor(random_tensor)


def _softmax(logits, dim=-1, axis=None, name=None):
    d = axis if axis is not None else dim
    z = logits - logits.max(dim=d, keepdim=True).values
    e = torch.exp(z)
    return e / e.sum(dim=d, keepdim=True)


def _elu(x, name=None):
    return torch.where(x > 0, x, torch.expm1(torch.minimum(x, torch.zeros_like(x))))


def _sparse_xent(labels=None, logits=None, name=None):
    z = logits - logits.max(dim=-1, keepdim=True).values
    lse = torch.log(torch.exp(z).sum(dim=-1))
    return lse - z.gather(-1, labels.to(torch.int64).unsqueeze(-1)).squeeze(-1)


def _conv2d(inp, filter=None, strides=None, padding="SAME", name=None):  # noqa: A002
    # NHWC input, HWIO filter; SAME padding with odd kernels and stride 1 is symmetric
    kh, kw = filter.shape[0], filter.shape[1]
    s = strides[1]
    if padding != "SAME" or s != 1 or kh % 2 == 0 or kw % 2 == 0:
        raise NotImplementedError("conv2d shim: SAME, stride 1, odd kernels only")
    out = torch.nn.functional.conv2d(inp.permute(0, 3, 1, 2), filter.permute(3, 2, 0, 1).contiguous(), padding=(kh // 2, kw // 2))
    return out.permute(0, 2, 3, 1)


nn.dropout = _dropout
nn.softmax = _softmax
nn.elu = _elu
nn.relu = lambda x, name=None: torch.relu(x)
nn.sigmoid = sigmoid
nn.tanh = tanh
nn.l2_loss = lambda x: (x ** 2).sum() / 2
nn.sparse_softmax_cross_entropy_with_logits = _sparse_xent
nn.conv2d = _conv2d

rnn_cell = types.ModuleType("tensorflow.nn.rnn_cell")


class RNNCell(object):
    """tf.nn.rnn_cell.RNNCell: the base class mac_cell.MACCell derives from (mac_cell.py:30)."""

    def zero_state(self, batch_size, dtype):
        raise NotImplementedError


class LSTMStateTuple(tuple):
    def __new__(cls, c, h):
        return tuple.__new__(cls, (c, h))

    c = property(lambda self: self[0])
    h = property(lambda self: self[1])


class BasicLSTMCell(RNNCell):
    """tf.nn.rnn_cell.BasicLSTMCell (rnn_cell_impl.py): variables `kernel` [in + n, 4 n] (get_variable's default glorot
    initialiser) and `bias` [4 n] (zeros) under <scope of the first call>/basic_lstm_cell; gates i, j, f, o =
    split([x, h] kernel + bias); c' = c sigmoid(f + forget_bias) + sigmoid(i) act(j); h' = act(c') sigmoid(o)."""

    def __init__(self, num_units, forget_bias=1.0, state_is_tuple=True, activation=None, reuse=None, name=None):
        self._n, self._fb, self._act, self._reuse = int(num_units), float(forget_bias), activation or torch.tanh, reuse
        self._vars = None

    state_size = property(lambda self: LSTMStateTuple(self._n, self._n))
    output_size = property(lambda self: self._n)

    def zero_state(self, batch_size, dtype):
        b = int(batch_size)
        return LSTMStateTuple(_w(torch.zeros((b, self._n), dtype=state.dtype)), _w(torch.zeros((b, self._n), dtype=state.dtype)))

    def __call__(self, inputs, st, scope=None):
        c, h = st
        if self._vars is None:          # Layer.__call__: the variables are built once, in the scope of the first call
            with variable_scope(scope or "basic_lstm_cell", reuse=self._reuse):
                kernel = get_variable("kernel", (int(inputs.shape[-1]) + self._n, 4 * self._n))
                bias = get_variable("bias", (4 * self._n,), zeros_initializer())
            self._vars = (kernel, bias)
        kernel, bias = self._vars
        z = torch.cat([inputs, h], dim=1) @ kernel + bias
        i, j, f, o = torch.split(z, self._n, dim=1)
        new_c = c * torch.sigmoid(f + self._fb) + torch.sigmoid(i) * self._act(j)
        new_h = self._act(new_c) * torch.sigmoid(o)
        return new_h, LSTMStateTuple(new_c, new_h)


def reverse_sequence(x, seq_lengths, seq_axis=1, batch_axis=0, seq_dim=None, batch_dim=None):
    """array_ops.reverse_sequence: the first seq_lengths[b] positions of row b reversed, the rest left where they are."""
    out = x.clone()
    for b in builtins.range(x.shape[0]):
        n = int(seq_lengths[b])
        out[b, :n] = torch.flip(x[b, :n], dims=[0])
    return _w(out)


def _dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, scope=None, **_ignored):
    """tf.nn.dynamic_rnn: variables under `scope or "rnn"`; past a sequence's end the output is zero and the state is copied
    through (rnn._rnn_step)."""
    if scope is None:
        with variable_scope("rnn"):
            return _dynamic_rnn(cell, inputs, sequence_length, initial_state, scope="rnn")
    B, T = inputs.shape[0], inputs.shape[1]
    st = initial_state
    outs = []
    for t in builtins.range(T):
        out, new = cell(inputs[:, t], st)
        if sequence_length is not None:
            live = (t < torch.as_tensor(sequence_length).reshape(B, 1)).to(out.dtype)
            out = live * out
            new = LSTMStateTuple(live * new[0] + (1 - live) * st[0], live * new[1] + (1 - live) * st[1])
        st = new
        outs.append(out)
    return _w(torch.stack(outs, dim=1)), st


def _bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, sequence_length=None, initial_state_fw=None, initial_state_bw=None,
                               **_ignored):
    """tf.nn.bidirectional_dynamic_rnn (rnn.py): scopes bidirectional_rnn/{fw,bw}; the backward cell runs over the
    length-reversed input and its outputs are reversed back."""
    with variable_scope("bidirectional_rnn"):
        with variable_scope("fw"):
            out_fw, st_fw = _dynamic_rnn(cell_fw, inputs, sequence_length, initial_state_fw, scope="fw")
        rev = reverse_sequence(inputs, sequence_length)
        with variable_scope("bw"):
            tmp, st_bw = _dynamic_rnn(cell_bw, rev, sequence_length, initial_state_bw, scope="bw")
        out_bw = reverse_sequence(tmp, sequence_length)
    return (out_fw, out_bw), (st_fw, st_bw)


def _embedding_lookup(params, ids, **_ignored):
    return _w(params[torch.as_tensor(ids).long()])


nn.dynamic_rnn = _dynamic_rnn
nn.bidirectional_dynamic_rnn = _bidirectional_dynamic_rnn
nn.embedding_lookup = _embedding_lookup
rnn_cell.RNNCell = RNNCell
rnn_cell.LSTMStateTuple = LSTMStateTuple
rnn_cell.BasicLSTMCell = BasicLSTMCell
for _n in ("BasicRNNCell", "GRUCell", "LSTMCell"):
    setattr(rnn_cell, _n, type(_n, (RNNCell,), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(
        NotImplementedError("the TF recurrent cells are outside the MAC-cell path and not shimmed"))}))
nn.rnn_cell = rnn_cell

# ------------------------------------------------------------------------------------------------
# tf.contrib
# ------------------------------------------------------------------------------------------------
contrib = types.ModuleType("tensorflow.contrib")
contrib.layers = types.ModuleType("tensorflow.contrib.layers")
contrib.layers.xavier_initializer = lambda uniform=True, seed=None, dtype=None: _Init("xavier")


def _batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True,
                updates_collections=None, scope=None, reuse=None, **_ignored):
    """tf.contrib.layers.batch_norm on the last axis.  Training: batch moments (biased variance) and, with
    updates_collections=None, an in-place moving-average update; inference: the moving moments."""
    with variable_scope(scope, default_name="BatchNorm", reuse=reuse):
        c = inputs.shape[-1]
        beta = get_variable("beta", (c,), zeros_initializer()) if center else None
        gamma = get_variable("gamma", (c,), ones_initializer()) if scale else None
        mm = get_variable("moving_mean", (c,), zeros_initializer(), trainable=False)
        mv = get_variable("moving_variance", (c,), ones_initializer(), trainable=False)
        if bool(is_training):
            axes = list(builtins.range(inputs.dim() - 1))
            mean = inputs.mean(dim=axes)
            var = ((inputs - mean) ** 2).mean(dim=axes)
            with torch.no_grad():
                torch.Tensor.sub_(mm, (1 - decay) * (mm - mean))
                torch.Tensor.sub_(mv, (1 - decay) * (mv - var))
        else:
            mean, var = mm, mv
        out = (inputs - mean) / torch.sqrt(var + epsilon)
        if gamma is not None:
            out = out * gamma
        if beta is not None:
            out = out + beta
    return out


contrib.layers.batch_norm = _batch_norm
contrib.rnn = types.ModuleType("tensorflow.contrib.rnn")
contrib.seq2seq = types.ModuleType("tensorflow.contrib.seq2seq")

# `import tensorflow as tf; tf.nn.rnn_cell...` only needs attributes, but register the submodules too
for _m in (nn, rnn_cell, contrib, contrib.layers, contrib.rnn, contrib.seq2seq):
    sys.modules[_m.__name__] = _m
