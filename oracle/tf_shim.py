"""A NumPy-backed stand-in for the handful of TensorFlow-1.x ops that the reference's SMPL code
uses -- TEST INFRASTRUCTURE ONLY.

TensorFlow 1.8 cannot be installed in this image, but `src/tf_smpl/{batch_smpl,batch_lbs,
projection}.py`, `src/omega.py` (OmegasPred) and the window logic of `src/evaluation/tester.py`
only call elementary ops (matmul, reshape, stack, concat, tile, pad, scatter_nd, cos, sin, norm,
...).  Installing this module as `tensorflow` lets those reference files be imported and EXECUTED
UNMODIFIED, in float64, to produce golden vectors (tests/golden/make_reference_golden.py).  That
pins the SMPL / projection / container / windowing rows of the path to the reference's own source
instead of to a restatement.  Every op below has exactly the documented TF semantics; nothing here
is specific to the reference.

The tf.contrib layers the reference calls (slim conv2d / batch_norm / max_pool2d / fully_connected /
arg_scope, contrib group_norm) are restated further down with their TF-1.8 semantics, and slim's
resnet_v2 / resnet_utils network definition is transcribed in oracle/slim_resnet_v2.py, so that the
reference's own encoder_resnet (src/models.py:50-77) can be executed as well.
"""
from __future__ import annotations

import contextlib
import sys
import types

import numpy as np

DTYPE = np.float64          # every tf.float32 of the reference is evaluated in this precision


class Dimension(object):
    def __init__(self, v):
        self.value = None if v is None else int(v)

    def __int__(self):
        return self.value

    __index__ = __int__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dimension) else o)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(self.value)

    # tf.Dimension arithmetic (src/omega.py computes B * T with T a Dimension)
    def __mul__(self, o): return Dimension(self.value * int(o))
    __rmul__ = __mul__
    def __add__(self, o): return Dimension(self.value + int(o))
    __radd__ = __add__
    def __sub__(self, o): return Dimension(self.value - int(o))
    def __rsub__(self, o): return Dimension(int(o) - self.value)
    def __floordiv__(self, o): return Dimension(self.value // int(o))

    def __repr__(self):
        return "Dimension(%r)" % self.value


class TensorShape(object):
    def __init__(self, dims):
        self.dims = [Dimension(d) for d in dims]

    def as_list(self):
        return [d.value for d in self.dims]

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __repr__(self):
        return "TensorShape(%r)" % self.as_list()


def _a(x):
    """Plain ndarray view of a Tensor / array-like (Dimensions become ints)."""
    if isinstance(x, Tensor):
        return x.a
    if isinstance(x, Dimension):
        return x.value
    if isinstance(x, (list, tuple)):
        if any(isinstance(e, (Tensor, Dimension)) for e in x):
            return np.asarray([_a(e) for e in x])
    return np.asarray(x)


def _ints(shape):
    return [int(_a(s)) for s in (shape if isinstance(shape, (list, tuple, TensorShape)) else [shape])]


class Tensor(object):
    """Eager value with the slice of the tf.Tensor API the reference touches."""
    __array_priority__ = 100

    def __init__(self, a):
        self.a = np.asarray(a)

    @property
    def shape(self):
        return TensorShape(self.a.shape)

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return self.a.dtype

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __getitem__(self, idx):
        return Tensor(self.a[idx])

    def __len__(self):
        return len(self.a)

    def _bin(self, o, f):
        return Tensor(f(self.a, _a(o)))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return Tensor(np.add(_a(o), self.a))
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return Tensor(np.subtract(_a(o), self.a))
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return Tensor(np.multiply(_a(o), self.a))
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return Tensor(np.divide(_a(o), self.a))
    __div__ = __truediv__
    def __neg__(self): return Tensor(-self.a)
    def __lt__(self, o): return self._bin(o, np.less)
    def __gt__(self, o): return self._bin(o, np.greater)

    def __repr__(self):
        return "Tensor(shape=%s)" % (self.a.shape,)


def _dt(dtype):
    if dtype is None:
        return None
    if dtype in (np.float32, np.float64, "float32", "float64") or dtype is float32:
        return DTYPE
    return dtype


float32 = np.dtype("float32").type
float64 = np.dtype("float64").type
int32 = np.dtype("int32").type


def _float_default(a, dtype):
    a = np.asarray(a)
    if dtype is not None:
        return a.astype(_dt(dtype))
    return a.astype(DTYPE) if a.dtype.kind == "f" else a


def Variable(initial_value, name=None, dtype=None, trainable=True, **kw):
    return Tensor(_float_default(_a(initial_value), dtype))


def constant(value, dtype=None, shape=None, name=None):
    a = np.asarray(_a(value))
    if shape is not None:
        shp = _ints(shape)
        a = np.zeros(shp, DTYPE) if a.size == 0 else np.broadcast_to(a, shp).copy()
    return Tensor(_float_default(a, dtype))


def reshape(x, shape, name=None):
    return Tensor(np.reshape(_a(x), _ints(shape)))


def shape(x, name=None):
    return list(_a(x).shape)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _a(a), _a(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return Tensor(np.matmul(a, b))


def stack(values, axis=0, name=None):
    return Tensor(np.stack([_a(v) for v in values], axis=axis))


def concat(values, axis, name=None):
    return Tensor(np.concatenate([_a(v) for v in values], axis=axis))


def tile(x, multiples, name=None):
    return Tensor(np.tile(_a(x), _ints(multiples)))


def expand_dims(x, axis=None, name=None, dim=None):
    return Tensor(np.expand_dims(_a(x), axis if axis is not None else dim))


def squeeze(x, axis=None, name=None):
    return Tensor(np.squeeze(_a(x), axis=None if axis is None else tuple(np.atleast_1d(axis))))


def eye(n, dtype=None, name=None):
    return Tensor(np.eye(int(_a(n)), dtype=_dt(dtype) or DTYPE))


def ones(shape, dtype=None, name=None):
    return Tensor(np.ones(_ints(shape), _dt(dtype) or DTYPE))


def zeros(shape, dtype=None, name=None):
    return Tensor(np.zeros(_ints(shape), _dt(dtype) or DTYPE))


def range(start, limit=None, delta=1, name=None):     # noqa: A001  (tf.range)
    if limit is None:
        start, limit = 0, start
    return Tensor(np.arange(int(_a(start)), int(_a(limit)), int(_a(delta)), dtype=np.int64))


def scatter_nd(indices, updates, shape, name=None):
    out = np.zeros(_ints(shape), dtype=_a(updates).dtype)
    idx = _a(indices)
    np.add.at(out, tuple(idx[..., i] for i in np.arange(idx.shape[-1])), _a(updates))   # duplicates accumulate
    return Tensor(out)


def norm(x, ord="euclidean", axis=None, keepdims=False, name=None, keep_dims=None):
    assert ord == "euclidean"
    x = _a(x)
    return Tensor(np.sqrt(np.sum(x * x, axis=axis, keepdims=bool(keepdims or keep_dims))))


def div(x, y, name=None):
    return Tensor(np.divide(_a(x), _a(y)))


def cos(x, name=None):
    return Tensor(np.cos(_a(x)))


def sin(x, name=None):
    return Tensor(np.sin(_a(x)))


def pad(x, paddings, mode="CONSTANT", name=None, constant_values=0):
    assert mode == "CONSTANT"
    return Tensor(np.pad(_a(x), [tuple(int(v) for v in p) for p in _a(paddings).tolist()], mode="constant",
                         constant_values=constant_values))


def gather(params, indices, axis=0, name=None):
    return Tensor(np.take(_a(params), _a(indices), axis=axis))


def cast(x, dtype, name=None):
    return Tensor(_a(x).astype(_dt(dtype)))


def stop_gradient(x, name=None):
    return x


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield name or default_name



# ------------------------------------------------------------------------------------------------
# Variable scopes + the three tf.contrib layers that src/models.py calls for f_movie and the IEF
# regressors.  The layer SEMANTICS below restate TF 1.8 (they live in TensorFlow, not in the
# reference); what running the reference through them pins is the reference's own WIRING: op order,
# residual connections, IEF recurrence, delta-omega assembly and -- through the scope stack -- the
# checkpoint variable names of SURVEY.md App. B.
# ------------------------------------------------------------------------------------------------
WEIGHTS = {}                # {checkpoint variable name: ndarray}, set by the golden generator
USED_VARIABLES = []         # names looked up, in order (lets the generator check the name contract)
_SCOPES = []
AUTO_REUSE = "AUTO_REUSE"


class _Scope(str):
    """What `with tf.variable_scope(...) as sc` binds: usable as the scope string, with the two attributes
    slim's resnet_v2 reads."""
    @property
    def name(self):
        return str(self)

    @property
    def original_name_scope(self):
        return str(self) + "/"


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, reuse=None, **kw):   # noqa: F811
    name = name_or_scope if name_or_scope is not None else default_name
    _SCOPES.append(str(name))
    try:
        yield _Scope("/".join(_SCOPES))
    finally:
        _SCOPES.pop()


# ---- slim.arg_scope / add_arg_scope (tensorflow/contrib/framework/python/ops/arg_scope.py) -------------
_ARG_SCOPES = [{}]          # stack of {function key: {kwarg: value}}


def _key(fn):
    return getattr(fn, "_key_op", None) or (fn.__module__, fn.__name__)


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    """`with arg_scope([ops], **defaults)` or `with arg_scope(saved_scope_dict)` (re-enter a scope)."""
    if isinstance(list_ops_or_scope, dict):
        assert not kwargs
        _ARG_SCOPES.append({k: dict(v) for k, v in list_ops_or_scope.items()})
        try:
            yield list_ops_or_scope
        finally:
            _ARG_SCOPES.pop()
        return
    cur = {k: dict(v) for k, v in _ARG_SCOPES[-1].items()}
    for op in list_ops_or_scope:
        assert hasattr(op, "_key_op"), "%r is not decorated with @add_arg_scope" % (op,)
        cur.setdefault(_key(op), {}).update(kwargs)
    _ARG_SCOPES.append(cur)
    try:
        yield cur
    finally:
        _ARG_SCOPES.pop()


def add_arg_scope(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        defaults = _ARG_SCOPES[-1].get(_key(wrapped))
        if defaults:
            merged = dict(defaults)
            merged.update(kwargs)          # explicit keyword arguments win over the scope's
            kwargs = merged
        return fn(*args, **kwargs)
    wrapped._key_op = (fn.__module__, fn.__name__)
    return wrapped


# ---- slim.utils: outputs collections (tensorflow/contrib/layers/python/layers/utils.py) ----------------
COLLECTIONS = {}            # {collection name: [(alias, Tensor)]}


class _Utils(object):
    @staticmethod
    def collect_named_outputs(collections, alias, outputs):
        if collections:
            COLLECTIONS.setdefault(collections, []).append((alias, outputs))
        return outputs

    @staticmethod
    def convert_collection_to_dict(collection, clear_collection=False):
        return dict(COLLECTIONS.get(collection, []))

    @staticmethod
    def last_dimension(shape, min_rank=1):
        dims = shape.as_list() if hasattr(shape, "as_list") else list(shape)
        assert len(dims) >= min_rank
        return dims[-1]


slim_utils = _Utils()


def _var(scope, leaf):
    name = "/".join(_SCOPES + [scope, leaf])
    USED_VARIABLES.append(name)
    if name not in WEIGHTS:
        raise KeyError("the reference asked for variable %r which the weight dict lacks" % name)
    return np.asarray(WEIGHTS[name], DTYPE)


def _relu(x, name=None):
    return Tensor(np.maximum(_a(x), 0))


def _fully_connected(inputs, num_outputs, activation_fn=_relu, weights_initializer=None, scope=None, **kw):
    """slim.fully_connected: activation_fn(inputs @ weights + biases); default activation ReLU."""
    w, b = _var(scope, "weights"), _var(scope, "biases")
    assert w.shape[1] == int(num_outputs)
    y = Tensor(np.matmul(_a(inputs), w) + b)
    return activation_fn(y) if activation_fn is not None else y


def _dropout(inputs, keep_prob=0.5, is_training=True, scope=None, **kw):
    assert not is_training, "inference only"
    return inputs


def _same_pads(size, k, s):
    """TF 'SAME': out = ceil(size / s); total pad = max((out-1)*s + k - size, 0), the smaller half first."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _pair(v):
    return [int(e) for e in v] if isinstance(v, (list, tuple)) else [int(v), int(v)]


@add_arg_scope
def _conv2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", data_format=None, rate=1,
            activation_fn=_relu, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
            weights_regularizer=None, biases_initializer="zeros", biases_regularizer=None, reuse=None,
            variables_collections=None, outputs_collections=None, trainable=True, scope=None, **kw):
    """tf.contrib.layers.conv2d == slim.conv2d (layers.py `convolution`), NHWC, rate 1:
    outputs = conv(inputs, weights) with 'SAME' or 'VALID' padding; then EITHER normalizer_fn(outputs,
    **normalizer_params) (no bias) OR + biases; then activation_fn.  Variables live under `scope`."""
    assert data_format in (None, "NHWC") and rate == 1 and padding in ("SAME", "VALID")
    kh, kw_ = _pair(kernel_size)
    sy, sx = _pair(stride)
    with variable_scope(scope, "Conv") as sc:
        w = _var_here("weights")                                    # HWIO
        x = _a(inputs)
        assert w.shape[:2] == (kh, kw_) and w.shape[2] == x.shape[3] and w.shape[3] == int(num_outputs), (w.shape, x.shape)
        H, W = x.shape[1], x.shape[2]
        if padding == "SAME":
            Ho, pt, pb = _same_pads(H, kh, sy)
            Wo, pl, pr = _same_pads(W, kw_, sx)
            xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        else:
            Ho, Wo, xp = (H - kh) // sy + 1, (W - kw_) // sx + 1, x
        y = np.zeros((x.shape[0], Ho, Wo, w.shape[3]), DTYPE)
        for i in np.arange(kh):
            for j in np.arange(kw_):
                y = y + np.matmul(xp[:, i:i + (Ho - 1) * sy + 1:sy, j:j + (Wo - 1) * sx + 1:sx, :], w[i, j])
        y = Tensor(y)
        if normalizer_fn is not None:
            y = normalizer_fn(y, **(normalizer_params or {}))
        elif biases_initializer is not None:
            y = Tensor(_a(y) + _var_here("biases"))
        if activation_fn is not None:
            y = activation_fn(y)
        return slim_utils.collect_named_outputs(outputs_collections, sc.name, y)


def _var_here(leaf):
    """Variable `leaf` of the innermost variable_scope."""
    name = "/".join(_SCOPES + [leaf])
    USED_VARIABLES.append(name)
    if name not in WEIGHTS:
        raise KeyError("the reference asked for variable %r which the weight dict lacks" % name)
    return np.asarray(WEIGHTS[name], DTYPE)


@add_arg_scope
def _batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None,
                param_initializers=None, updates_collections=None, is_training=True, reuse=None,
                variables_collections=None, outputs_collections=None, trainable=True, scope=None, **kw):
    """slim.batch_norm (layers.py `batch_norm`) in inference mode: is_training=False normalises with the
    moving statistics: gamma * (x - moving_mean) * rsqrt(moving_variance + epsilon) + beta; default scope
    'BatchNorm'; `gamma` exists only with scale=True."""
    assert not is_training, "inference only"
    with variable_scope(scope, "BatchNorm") as sc:
        x = _a(inputs)
        beta = _var_here("beta") if center else 0.0
        gamma = _var_here("gamma") if scale else 1.0
        mean, var = _var_here("moving_mean"), _var_here("moving_variance")
        y = Tensor((x - mean) / np.sqrt(var + epsilon) * gamma + beta)
        if activation_fn is not None:
            y = activation_fn(y)
        return slim_utils.collect_named_outputs(outputs_collections, sc.name, y)


@add_arg_scope
def _max_pool2d(inputs, kernel_size, stride=2, padding="VALID", data_format=None, outputs_collections=None,
                scope=None):
    """slim.max_pool2d = nn.max_pool; 'SAME' pads like _same_pads and padded positions never win."""
    assert data_format in (None, "NHWC") and padding in ("SAME", "VALID")
    kh, kw_ = _pair(kernel_size)
    sy, sx = _pair(stride)
    x = _a(inputs)
    H, W = x.shape[1], x.shape[2]
    if padding == "SAME":
        Ho, pt, pb = _same_pads(H, kh, sy)
        Wo, pl, pr = _same_pads(W, kw_, sx)
        xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    else:
        Ho, Wo, xp = (H - kh) // sy + 1, (W - kw_) // sx + 1, x
    y = np.full((x.shape[0], Ho, Wo, x.shape[3]), -np.inf, DTYPE)
    for i in np.arange(kh):
        for j in np.arange(kw_):
            y = np.maximum(y, xp[:, i:i + (Ho - 1) * sy + 1:sy, j:j + (Wo - 1) * sx + 1:sx, :])
    return slim_utils.collect_named_outputs(outputs_collections, scope, Tensor(y))


def reduce_mean(x, axis=None, keepdims=None, name=None, keep_dims=None):
    ax = None if axis is None else tuple(int(a) for a in np.atleast_1d(_a(axis)))
    return Tensor(np.mean(_a(x), axis=ax, keepdims=bool(keepdims or keep_dims)))


def _group_norm(inputs, groups=32, channels_axis=-1, reduction_axes=(-3, -2), center=True, scale=True,
                epsilon=1e-6, activation_fn=None, scope=None, reuse=None, **kw):
    """tf.contrib.layers.group_norm (TF 1.8): moments over reduction_axes and the within-group
    channel sub-axis, population variance; gain = rsqrt(var+eps)*gamma; offset = -mean*gain+beta."""
    x = _a(inputs)
    nd = x.ndim
    ca = channels_axis % nd
    assert ca == nd - 1, "channels-last only"
    C = x.shape[ca]
    xg = x.reshape(x.shape[:-1] + (groups, C // groups))
    axes = tuple(a % nd for a in reduction_axes) + (nd,)             # reduction axes + channel sub-axis
    mean = xg.mean(axis=axes, keepdims=True)
    var = ((xg - mean) ** 2).mean(axis=axes, keepdims=True)
    gamma = _var(scope, "gamma").reshape((1,) * (nd - 1) + (groups, C // groups)) if scale else 1.0
    beta = _var(scope, "beta").reshape((1,) * (nd - 1) + (groups, C // groups)) if center else 0.0
    gain = gamma / np.sqrt(var + epsilon)
    offset = beta - mean * gain
    y = Tensor((xg * gain + offset).reshape(x.shape))
    return activation_fn(y) if activation_fn is not None else y


def add(x, y, name=None):
    return Tensor(np.add(_a(x), _a(y)))


class _NN(object):
    relu = staticmethod(_relu)


nn = _NN()


def install(precision=np.float64):
    """Register this module as `tensorflow` (and inert stubs for the other imports of the reference
    files) in sys.modules.  Returns the list of names it added so the caller can remove them."""
    global DTYPE
    DTYPE = precision
    me = sys.modules[__name__]
    added = []

    def put(name, mod):
        if name not in sys.modules:
            sys.modules[name] = mod
            added.append(name)

    put("tensorflow", me)
    contrib = types.ModuleType("tensorflow.contrib")
    put("tensorflow.contrib", contrib)
    for sub in ("tensorflow.contrib.slim", "tensorflow.contrib.layers", "tensorflow.contrib.layers.python",
                "tensorflow.contrib.layers.python.layers", "tensorflow.contrib.layers.python.layers.initializers",
                "tensorflow.contrib.framework", "tensorflow.contrib.slim.python", "tensorflow.contrib.slim.python.slim",
                "tensorflow.contrib.slim.python.slim.nets"):
        m = types.ModuleType(sub)
        m.variance_scaling_initializer = lambda *a, **k: None
        m.l2_regularizer = lambda *a, **k: None
        m.fully_connected, m.dropout = _fully_connected, _dropout
        m.conv2d, m.group_norm = _conv2d, _group_norm
        m.batch_norm, m.max_pool2d = _batch_norm, _max_pool2d
        m.arg_scope, m.add_arg_scope = arg_scope, add_arg_scope
        m.utils = slim_utils
        m.get_variables = lambda *a, **k: []
        put(sub, m)
        setattr(contrib, sub.split(".")[2], sys.modules[sub]) if sub.count(".") == 2 else None
    me.contrib = contrib
    # slim's network definition (tf.contrib.slim.python.slim.nets.{resnet_utils,resnet_v2}), transcribed
    from oracle import slim_resnet_v2
    nets = sys.modules["tensorflow.contrib.slim.python.slim.nets"]
    nets.resnet_v2, nets.resnet_utils = slim_resnet_v2.resnet_v2_module, slim_resnet_v2.resnet_utils
    put("tensorflow.contrib.slim.python.slim.nets.resnet_v2", slim_resnet_v2.resnet_v2_module)
    put("tensorflow.contrib.slim.python.slim.nets.resnet_utils", slim_resnet_v2.resnet_utils)
    for stub in ("deepdish", "ipdb", "cv2"):
        put(stub, types.ModuleType(stub))
    return added


def uninstall(added):
    for name in added:
        sys.modules.pop(name, None)
