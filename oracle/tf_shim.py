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

Not covered (so those rows stay pinned by the restated oracle only): tf.contrib.slim's
resnet_v2_50, tf.contrib.layers.group_norm / conv2d, slim.fully_connected.
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


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, reuse=None, **kw):   # noqa: F811
    name = name_or_scope if name_or_scope is not None else default_name
    _SCOPES.append(str(name))
    try:
        yield "/".join(_SCOPES)
    finally:
        _SCOPES.pop()


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


def _conv2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", data_format="NHWC", rate=1,
            activation_fn=_relu, weights_initializer=None, scope=None, reuse=None, **kw):
    """tf.contrib.layers.conv2d on NHWC with stride 1 / rate 1: SAME zero padding, + bias."""
    assert padding == "SAME" and data_format == "NHWC" and stride == 1 and rate == 1
    w, b = _var(scope, "weights"), _var(scope, "biases")            # HWIO
    x = _a(inputs)
    kh, kw_, cin, cout = w.shape
    assert [kh, kw_] == [int(k) for k in kernel_size] and cout == int(num_outputs)
    ph, pw = kh - 1, kw_ - 1
    xp = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    H, W = x.shape[1], x.shape[2]
    y = np.zeros(x.shape[:3] + (cout,), DTYPE) + b
    for i in np.arange(kh):
        for j in np.arange(kw_):
            y = y + np.matmul(xp[:, i:i + H, j:j + W, :], w[i, j])
    y = Tensor(y)
    return activation_fn(y) if activation_fn is not None else y


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
                "tensorflow.contrib.framework"):
        m = types.ModuleType(sub)
        m.variance_scaling_initializer = lambda *a, **k: None
        m.fully_connected, m.dropout = _fully_connected, _dropout
        m.conv2d, m.group_norm = _conv2d, _group_norm
        m.get_variables = lambda *a, **k: []
        put(sub, m)
        setattr(contrib, sub.split(".")[2], sys.modules[sub]) if sub.count(".") == 2 else None
    me.contrib = contrib
    for stub in ("deepdish", "ipdb", "cv2"):
        put(stub, types.ModuleType(stub))
    return added


def uninstall(added):
    for name in added:
        sys.modules.pop(name, None)
