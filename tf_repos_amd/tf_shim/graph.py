"""A minimal TF-1.x graph-mode front end: `tf.*` calls made by the reference's model_fn / input_fn build a symbolic
graph of `Tensor` nodes instead of running anything.  The graph is then LOWERED (lowering.py) onto the HIP engine --
nothing here computes; there is no interpreter and no CPU fallback.  Only the symbols the reference scripts touch are
provided (SURVEY 8b census); anything else raises AttributeError / NotImplementedError loudly.
"""
from __future__ import annotations

import itertools
import threading
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import errors


class DType:
    def __init__(self, name, np_dtype):
        self.name, self.np = name, np_dtype

    def __repr__(self):
        return "tf." + self.name

    @property
    def as_numpy_dtype(self):
        return self.np


float32 = DType("float32", np.float32)
int32 = DType("int32", np.int32)
int64 = DType("int64", np.int64)
bool_ = DType("bool", np.bool_)
string = DType("string", object)

_uid = itertools.count()


class _Scope(threading.local):
    def __init__(self):
        self.stack: List[str] = []
        self.graph: Optional["Graph"] = None


_scope = _Scope()


class Graph:
    def __init__(self):
        self.nodes: List["Tensor"] = []
        self.variables: Dict[str, "Variable"] = {}
        self.collections: Dict[str, list] = {}
        self.global_step: Optional["Variable"] = None

    def __enter__(self):
        self._prev = _scope.graph
        _scope.graph = self
        _scope.stack = []
        return self

    def __exit__(self, *exc):
        _scope.graph = self._prev
        return False


def current_graph() -> Graph:
    if _scope.graph is None:
        _scope.graph = Graph()
    return _scope.graph


def scope_prefix() -> str:
    return "/".join(_scope.stack) + ("/" if _scope.stack else "")


class variable_scope:
    def __init__(self, name, reuse=None, **_kw):
        self.name = name

    def __enter__(self):
        _scope.stack.append(self.name)
        return self

    def __exit__(self, *exc):
        _scope.stack.pop()
        return False


class name_scope:
    """tf.name_scope names ops only: variables created under it keep their variable_scope name [TF-1.x]
    (DeepCvrMTL.py:166,185: 'cvr_mlp0/weights', not 'CVR_Task/cvr_mlp0/weights')."""

    def __init__(self, name, *a, **_kw):
        self.name = name

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


AUTO_REUSE = "AUTO_REUSE"


def _bshape(a, b):
    """numpy-style broadcast of static shapes with None wildcards."""
    a, b = list(a or ()), list(b or ())
    n = max(len(a), len(b))
    a, b = [1] * (n - len(a)) + a, [1] * (n - len(b)) + b
    out = []
    for x, y in zip(a, b):
        if x == 1:
            out.append(y)
        elif y == 1 or y == x:
            out.append(x)
        elif x is None or y is None:
            out.append(x if y is None else y)
        else:
            raise ValueError("Dimensions must be equal, but are %s and %s" % (x, y))
    return tuple(out)


class Tensor:
    def __init__(self, op: str, inputs: Sequence[Any] = (), attrs: Optional[Dict[str, Any]] = None, dtype: DType = float32,
                 shape: Optional[Tuple] = None, name: Optional[str] = None):
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.dtype = dtype
        self.shape = tuple(shape) if shape is not None else None
        self.id = next(_uid)
        self.scope = scope_prefix()
        self.name = name or "%s%s_%d" % (self.scope, op, self.id)
        current_graph().nodes.append(self)

    # TF-style accessors the scripts use
    def get_shape(self):
        return self.shape

    # python operators -> graph ops
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return subtract(self, o)
    def __rsub__(self, o): return subtract(o, self)
    def __mul__(self, o): return multiply(self, o)
    def __rmul__(self, o): return multiply(o, self)
    def __neg__(self): return multiply(-1.0, self)
    def __gt__(self, o): return Tensor("greater", [self, _t(o, self.dtype)], {}, bool_, self.shape)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        shp = list(self.shape) if self.shape is not None else None
        out = None
        if shp is not None:
            out = []
            for i, k in enumerate(key):
                if isinstance(k, slice):
                    if k == slice(None):
                        out.append(shp[i])
                    else:
                        n = shp[i]
                        out.append(None if n is None else len(range(*k.indices(n))))
                # an int index drops the dimension
            out.extend(shp[len(key):])
            out = tuple(out)
        return Tensor("getitem", [self], {"key": key}, self.dtype, out)

    def __repr__(self):
        return "<tf.Tensor '%s' op=%s shape=%s dtype=%s>" % (self.name, self.op, self.shape, self.dtype.name)

    def __bool__(self):
        raise TypeError("Using a tf.Tensor as a Python bool is not allowed (graph mode).")

    def __iter__(self):
        raise TypeError("Tensor objects are not iterable (graph mode).")


class Variable(Tensor):
    def __init__(self, name, shape, initializer, dtype=float32, trainable=True):
        super().__init__("variable", [], {"initializer": initializer, "trainable": trainable}, dtype, tuple(int(s) for s in shape), name)
        self.var_name = name
        self.initializer = initializer
        self.trainable = trainable


class Initializer:
    def __init__(self, kind, **kw):
        self.kind, self.kw = kind, kw


def glorot_normal_initializer(seed=None, dtype=float32):
    return Initializer("glorot_normal", seed=seed)


def glorot_uniform_initializer(seed=None, dtype=float32):
    return Initializer("glorot_uniform", seed=seed)


def constant_initializer(value=0.0, dtype=float32):
    return Initializer("constant", value=float(value))


def zeros_initializer(dtype=float32):
    return Initializer("constant", value=0.0)


def ones_initializer(dtype=float32):
    return Initializer("constant", value=1.0)


def get_variable(name, shape=None, dtype=float32, initializer=None, trainable=True, **_kw):
    g = current_graph()
    full = scope_prefix() + name
    if full in g.variables:
        return g.variables[full]
    if shape is None:
        raise ValueError("Shape of a new variable (%s) must be fully defined" % full)
    v = Variable(full, shape, initializer or glorot_uniform_initializer(), dtype, trainable)
    g.variables[full] = v
    return v


def _t(x, dtype=float32):
    if isinstance(x, Tensor):
        return x
    a = np.asarray(x)
    return Tensor("const", [], {"value": a}, dtype if a.dtype.kind == "f" or dtype is not float32 else (int32 if a.dtype.kind in "iu" else float32),
                  a.shape)


def constant(value, dtype=None, shape=None, name=None):
    a = np.asarray(value)
    dt = dtype or (float32 if a.dtype.kind == "f" else int32)
    return Tensor("const", [], {"value": a}, dt, a.shape if shape is None else tuple(shape), name)


def placeholder(dtype, shape=None, name=None):
    return Tensor("placeholder", [], {}, dtype, tuple(shape) if shape is not None else None, name)


# ---- math ---------------------------------------------------------------------------------------------------------
def _binary(op, a, b):
    a, b = _t(a), _t(b)
    return Tensor(op, [a, b], {}, a.dtype, _bshape(a.shape, b.shape) if a.shape is not None and b.shape is not None else None)


def add(a, b, name=None): return _binary("add", a, b)
def subtract(a, b, name=None): return _binary("sub", a, b)
def multiply(a, b, name=None): return _binary("mul", a, b)


def square(x, name=None):
    return Tensor("square", [x], {}, x.dtype, x.shape)


def sigmoid(x, name=None):
    return Tensor("sigmoid", [x], {}, x.dtype, x.shape)


def identity(x, name=None):
    return Tensor("identity", [x], {}, x.dtype, x.shape)


def ones_like(x, dtype=None, name=None):
    return Tensor("ones_like", [x], {}, dtype or x.dtype, x.shape)


def cast(x, dtype, name=None):
    if not isinstance(x, Tensor):
        return Tensor("const", [], {"value": np.asarray(x)}, dtype, ())
    return Tensor("cast", [x], {}, dtype, x.shape)


def cond(pred, true_fn, false_fn, name=None):
    p = pred
    while isinstance(p, Tensor) and p.op == "cast":
        p = p.inputs[0]
    if isinstance(p, Tensor) and p.op == "const":
        return true_fn() if bool(p.attrs["value"]) else false_fn()
    raise NotImplementedError("tf.cond on a non-constant predicate")


def _reduce(op, x, axis, keep_dims=False):
    shp = None
    if x.shape is not None:
        axes = list(range(len(x.shape))) if axis is None else ([axis] if isinstance(axis, int) else list(axis))
        axes = [a % len(x.shape) for a in axes]
        shp = tuple(s for i, s in enumerate(x.shape) if i not in axes)
    return Tensor(op, [x], {"axis": axis}, x.dtype, shp)


def reduce_sum(x, axis=None, keep_dims=False, name=None, reduction_indices=None):
    return _reduce("reduce_sum", x, axis if axis is not None else reduction_indices)


def reduce_mean(x, axis=None, keep_dims=False, name=None):
    return _reduce("reduce_mean", x, axis)


def reshape(x, shape, name=None):
    if isinstance(shape, Tensor):       # reshape(values, dense_shape) in decode_libsvm
        return Tensor("reshape", [x, shape], {"shape": None}, x.dtype, None)
    dyn = [s for s in shape if isinstance(s, Tensor)]            # e.g. [-1, tf.shape(ids)[1], 1] (DIN.py:169): dynamic dims
    if dyn:
        out = tuple(None if (isinstance(s, Tensor) or int(s) == -1) else int(s) for s in shape)
        return Tensor("reshape", [x] + dyn, {"shape": tuple(None if isinstance(s, Tensor) else int(s) for s in shape)}, x.dtype, out)
    shape = [int(s) for s in shape]
    out = list(shape)
    if x.shape is not None and all(s is not None for s in x.shape) and -1 in out:
        tot = int(np.prod(x.shape))
        known = int(np.prod([s for s in out if s != -1]))
        out[out.index(-1)] = tot // known if known else None
    out = tuple(None if s == -1 else s for s in out)
    return Tensor("reshape", [x], {"shape": tuple(shape)}, x.dtype, out)


def matmul(a, b, name=None, **_kw):
    shp = None
    if a.shape is not None and b.shape is not None:
        shp = (a.shape[0], b.shape[1])
    return Tensor("matmul", [a, b], {}, a.dtype, shp)


def concat(values, axis, name=None):
    shp = None
    if all(v.shape is not None for v in values):
        shp = list(values[0].shape)
        dims = [v.shape[axis] for v in values]
        shp[axis] = None if any(d is None for d in dims) else sum(dims)
        shp = tuple(shp)
    return Tensor("concat", list(values), {"axis": axis}, values[0].dtype, shp)


def stack(values, axis=0, name=None):
    values = list(values)
    shp = None
    if values and values[0].shape is not None:
        shp = list(values[0].shape)
        shp.insert(axis, len(values))
        shp = tuple(shp)
    return Tensor("stack", values, {"axis": axis}, values[0].dtype, shp)


def transpose(x, perm=None, name=None):
    shp = tuple(x.shape[p] for p in perm) if x.shape is not None and perm is not None else None
    return Tensor("transpose", [x], {"perm": tuple(perm) if perm is not None else None}, x.dtype, shp)


def gather(params, indices, axis=0, name=None):
    idx = list(indices) if not isinstance(indices, Tensor) else indices
    shp = None
    if params.shape is not None and not isinstance(idx, Tensor):
        shp = list(params.shape)
        shp[axis] = len(idx)
        shp = tuple(shp)
    return Tensor("gather", [params] + ([idx] if isinstance(idx, Tensor) else []),
                  {"axis": axis, "indices": None if isinstance(idx, Tensor) else tuple(int(i) for i in idx)}, params.dtype, shp)


def einsum(equation, *inputs, **_kw):
    shp = None
    if equation.replace(" ", "") == "api,apj->apij" and all(i.shape is not None for i in inputs):
        a, p, i = inputs[0].shape
        shp = (a, p, i, inputs[1].shape[2])
    return Tensor("einsum", list(inputs), {"equation": equation.replace(" ", "")}, inputs[0].dtype, shp)


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    n = int(num_or_size_splits)
    outs = []
    for i in range(n):
        shp = None
        if value.shape is not None:
            shp = list(value.shape)
            shp[axis] = None if shp[axis] is None else shp[axis] // n
            shp = tuple(shp)
        outs.append(Tensor("split", [value], {"axis": axis, "num": n, "index": i}, value.dtype, shp))
    return outs


# ---- strings (decode_libsvm, DeepFM.py:69-75) --------------------------------------------------------------------------
class SparseStrings:
    """What tf.string_split returns: .values / .dense_shape (symbolic)."""

    def __init__(self, node):
        self.node = node
        self.values = Tensor("sparse_values", [node], {}, string, (None,))
        self.dense_shape = Tensor("sparse_dense_shape", [node], {}, int64, (2,))
        self.indices = Tensor("sparse_indices", [node], {}, int64, (None, 2))


def string_split(source, delimiter=" ", skip_empty=True):
    if isinstance(source, (list, tuple)):
        source = stack([s if isinstance(s, Tensor) else constant(s, string) for s in source]) if len(source) != 1 else \
            Tensor("pack1", [source[0]], {}, string, (1,))
    node = Tensor("string_split", [source], {"delimiter": delimiter, "skip_empty": skip_empty}, string, None)
    return SparseStrings(node)


def string_to_number(string_tensor, out_type=float32, name=None):
    return Tensor("string_to_number", [string_tensor], {"out_type": out_type}, out_type, string_tensor.shape)


def decode_csv(records, record_defaults, field_delim=",", **_kw):
    return [Tensor("decode_csv", [records], {"index": i, "default": d, "n": len(record_defaults)},
                   float32 if isinstance(d[0], float) else int32, ()) for i, d in enumerate(record_defaults)]


# ---- nn / layers ------------------------------------------------------------------------------------------------------
def embedding_lookup(params, ids, name=None, **_kw):
    shp = None
    if ids.shape is not None and params.shape is not None:
        shp = tuple(ids.shape) + tuple(params.shape[1:])
    return Tensor("embedding_lookup", [params, ids], {}, params.dtype, shp)


def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    return Tensor("dropout", [x], {"keep_prob": float(keep_prob)}, x.dtype, x.shape)


def softmax(logits, dim=-1, name=None, axis=None):
    return Tensor("softmax", [logits], {"axis": dim if axis is None else axis}, logits.dtype, logits.shape)


def l2_loss(t, name=None):
    return Tensor("l2_loss", [t], {}, t.dtype, ())


def sigmoid_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, name=None):
    if _sentinel is not None:
        raise ValueError("Only call `sigmoid_cross_entropy_with_logits` with named arguments (labels=..., logits=...)")
    return Tensor("sigmoid_xent", [logits, labels], {}, logits.dtype, logits.shape)


def relu(x, name=None):
    return Tensor("relu", [x], {}, x.dtype, x.shape)


class _Regularizer:
    def __init__(self, scale):
        self.scale = scale


def l2_regularizer(scale, scope=None):
    return _Regularizer(float(scale))


def fully_connected(inputs, num_outputs, activation_fn=relu, normalizer_fn=None, normalizer_params=None,
                    weights_initializer=None, weights_regularizer=None, biases_initializer=None, scope=None, reuse=None, **_kw):
    """contrib.layers.fully_connected: relu unless overridden, weights [in,out] Xavier-uniform, biases zeros [TF-1.4].
    weights_regularizer only feeds REGULARIZATION_LOSSES, which the reference's loss never reads."""
    if normalizer_fn is not None:
        raise NotImplementedError("fully_connected(normalizer_fn=...) is not used by the reference and not supported")
    in_dim = inputs.shape[-1] if inputs.shape is not None else None
    if in_dim is None:
        raise ValueError("The last dimension of the inputs to `fully_connected` should be defined. Found `None`.")
    with variable_scope(scope or "fully_connected"):
        w = get_variable("weights", [in_dim, num_outputs], initializer=weights_initializer or glorot_uniform_initializer())
        b = get_variable("biases", [num_outputs], initializer=biases_initializer or zeros_initializer())
        if weights_regularizer is not None:
            current_graph().collections.setdefault("regularization_losses", []).append((w.var_name, weights_regularizer.scale))
    act = "relu" if activation_fn is relu else ("identity" if activation_fn in (identity, None) else ("sigmoid" if activation_fn is sigmoid else None))
    if act is None:
        raise NotImplementedError("fully_connected activation %r" % activation_fn)
    shp = tuple(inputs.shape[:-1]) + (num_outputs,)
    return Tensor("fully_connected", [inputs, w, b], {"activation": act, "num_outputs": int(num_outputs)}, inputs.dtype, shp)


def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, updates_collections="update_ops",
               is_training=True, reuse=None, scope=None, **_kw):
    n = inputs.shape[-1]
    with variable_scope(scope or "BatchNorm"):
        beta = get_variable("beta", [n], initializer=zeros_initializer()) if center else None
        gamma = get_variable("gamma", [n], initializer=ones_initializer()) if scale else None
        mm = get_variable("moving_mean", [n], initializer=zeros_initializer(), trainable=False)
        mv = get_variable("moving_variance", [n], initializer=ones_initializer(), trainable=False)
    ins = [inputs] + [v for v in (beta, gamma, mm, mv) if v is not None]
    return Tensor("batch_norm", ins, {"decay": decay, "epsilon": epsilon, "is_training": bool(is_training), "center": center,
                                       "scale": scale}, inputs.dtype, inputs.shape)


# ---- tf.Example parsing + variable-length lookups (DIN.py:57-97,143-183; DeepCvrMTL.py:61-104,153-165) ------------------------
class FixedLenFeature:
    def __init__(self, shape, dtype, default_value=None):
        self.shape, self.dtype, self.default_value = list(shape), dtype, default_value


class VarLenFeature:
    def __init__(self, dtype):
        self.dtype = dtype


def parse_single_example(serialized, features, name=None, example_names=None):
    """-> {key: Tensor}; FixedLenFeature -> dense tensor of its shape, VarLenFeature -> a SparseTensor node."""
    out = {}
    for key, spec in features.items():
        if isinstance(spec, FixedLenFeature):
            if spec.default_value is not None:
                raise errors.UnimplementedError("FixedLenFeature(default_value=...) is not used by the reference scripts")
            out[key] = Tensor("parsed_fixed", [serialized], {"key": key, "shape": tuple(spec.shape)}, spec.dtype, tuple(spec.shape))
        elif isinstance(spec, VarLenFeature):
            out[key] = Tensor("parsed_varlen", [serialized], {"key": key}, spec.dtype, (None,))
        else:
            raise errors.UnimplementedError("parse_single_example feature spec %r" % (spec,))
    return out


def embedding_lookup_sparse(params, sp_ids, sp_weights, partition_strategy="mod", name=None, combiner=None, max_norm=None):
    """combiner="sum": out[b] = sum_j w_j params[id_j] over row b's entries (DIN.py:148,180-183)."""
    if combiner != "sum":
        raise errors.UnimplementedError("embedding_lookup_sparse(combiner=%r): the reference scripts use 'sum'" % (combiner,))
    ins = [params, sp_ids] + ([sp_weights] if sp_weights is not None else [])
    return Tensor("embedding_lookup_sparse", ins, {"weighted": sp_weights is not None}, params.dtype, (None, params.shape[1]))


def sparse_tensor_to_dense(sp_input, default_value=0, validate_indices=True, name=None):
    """[B, P] dense form of a parsed VarLenFeature, rows padded with default_value up to the batch's longest (DIN.py:153-154)."""
    if not (isinstance(sp_input, Tensor) and sp_input.op == "iterator_varlen"):
        raise errors.UnimplementedError("sparse_tensor_to_dense of something that is not a parsed VarLenFeature")
    if default_value != 0:
        raise errors.UnimplementedError("sparse_tensor_to_dense(default_value != 0)")
    return Tensor("sparse_to_dense", [sp_input], {}, sp_input.dtype, (None, None))


def expand_dims(x, axis=None, name=None, dim=None):
    axis = dim if axis is None else axis
    shp = None
    if x.shape is not None:
        shp = list(x.shape)
        shp.insert(axis % (len(shp) + 1) if axis < 0 else axis, 1)
        shp = tuple(shp)
    return Tensor("expand_dims", [x], {"axis": axis}, x.dtype, shp)


def shape(x, name=None, out_type=None):
    return Tensor("shape", [x], {}, int32, (len(x.shape),) if x.shape is not None else None)


def tile(x, multiples, name=None):
    dyn = [m for m in multiples if isinstance(m, Tensor)]
    shp = None
    if x.shape is not None:
        shp = tuple(None if (isinstance(m, Tensor) or d is None) else d * int(m) for d, m in zip(x.shape, multiples))
    return Tensor("tile", [x] + dyn, {"multiples": tuple(None if isinstance(m, Tensor) else int(m) for m in multiples)}, x.dtype, shp)


def log_loss(labels, predictions, weights=1.0, epsilon=1e-7, scope=None, **_kw):
    """tf.losses.log_loss: mean over the batch of -z log(p+eps) - (1-z) log(1-p+eps) [TF-1.4] (DeepCvrMTL.py:224)."""
    if weights != 1.0:
        raise errors.UnimplementedError("log_loss(weights=...)")
    return Tensor("log_loss", [_t(labels), _t(predictions)], {"epsilon": float(epsilon)}, float32, ())


def summary_scalar(name, tensor, **_kw):
    return None


# ---- metrics / train ------------------------------------------------------------------------------------------------------
def metrics_auc(labels, predictions, num_thresholds=200, **_kw):
    t = Tensor("metrics_auc", [labels, predictions], {"num_thresholds": num_thresholds}, float32, ())
    return (t, t)


def get_global_step(graph=None):
    g = current_graph()
    if g.global_step is None:
        g.global_step = Variable("global_step", (), constant_initializer(0), int64, trainable=False)
        g.variables["global_step"] = g.global_step
    return g.global_step


get_or_create_global_step = get_global_step


class Optimizer:
    kind = None

    def __init__(self, learning_rate, **hyper):
        self.learning_rate = float(learning_rate)
        self.hyper = hyper

    def minimize(self, loss, global_step=None, var_list=None, **_kw):
        return Tensor("minimize", [loss], {"optimizer": self.kind, "learning_rate": self.learning_rate, "hyper": dict(self.hyper),
                                           "global_step": global_step is not None}, float32, ())


class AdamOptimizer(Optimizer):
    kind = "Adam"

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08, **_kw):
        super().__init__(learning_rate, beta1=beta1, beta2=beta2, epsilon=epsilon)


class AdagradOptimizer(Optimizer):
    kind = "Adagrad"

    def __init__(self, learning_rate, initial_accumulator_value=0.1, **_kw):
        super().__init__(learning_rate, initial_accumulator_value=initial_accumulator_value)


class MomentumOptimizer(Optimizer):
    kind = "Momentum"

    def __init__(self, learning_rate, momentum, use_nesterov=False, **_kw):
        super().__init__(learning_rate, momentum=momentum, use_nesterov=use_nesterov)


class FtrlOptimizer(Optimizer):
    kind = "ftrl"

    def __init__(self, learning_rate, learning_rate_power=-0.5, initial_accumulator_value=0.1, l1_regularization_strength=0.0,
                 l2_regularization_strength=0.0, **_kw):
        super().__init__(learning_rate, learning_rate_power=learning_rate_power, initial_accumulator_value=initial_accumulator_value,
                         l1=l1_regularization_strength, l2=l2_regularization_strength)


def ancestors(roots: Sequence[Tensor]) -> List[Tensor]:
    """All nodes reachable from roots (inputs direction), topologically ordered by creation id."""
    seen, out, stack = set(), [], [r for r in roots if isinstance(r, Tensor)]
    while stack:
        n = stack.pop()
        if n.id in seen:
            continue
        seen.add(n.id)
        out.append(n)
        for i in n.inputs:
            if isinstance(i, Tensor):
                stack.append(i)
    return sorted(out, key=lambda t: t.id)
