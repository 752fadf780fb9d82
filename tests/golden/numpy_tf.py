"""A deferred-NumPy stand-in for the TensorFlow 1.x ops the reference requests (build container only).

TensorFlow, gpflow and ``future`` are not in the image, so the reference cannot be imported as it
is.  Its Python code, however, only BUILDS small op graphs (``tf.placeholder`` ... ``tf.matmul``
... ``tf.less``) and later evaluates them (``.eval(feed_dict)`` / ``session.run``).  This module
answers every op the reference's ``functions.py`` / ``lyapunov.py`` / ``reinforcement_learning.py``
/ ``examples/utilities.py`` request on the hot path with a node that remembers the NumPy function
of the same meaning; evaluation runs NumPy.  Nothing of the reference is copied: the reference's
files are loaded from ``/root/reference`` and executed unmodified by the fixture generators
(``make_reference_safe_sets.py``, ``make_reference_policy_iteration.py``,
``make_reference_functions.py``).

What this is and is not:

* Elementwise arithmetic, comparisons, ``sin`` / ``cos`` / ``tanh``, clipping, gathers, splits and
  concatenations are IEEE-754 double operations with one rounding each in TensorFlow and in NumPy
  alike (``sin`` / ``cos`` / ``tanh`` may differ by an ulp between the two libraries' kernels).
* ``tf.matmul`` and ``tf.reduce_sum`` / ``tf.norm`` accumulate LEFT TO RIGHT, one rounding per
  multiply and per add, no fused multiply-add: TensorFlow's Eigen kernels (and BLAS) do not
  define an order, the product's contract does - this is the order of the
  oracle and of the HIP kernels, restated here independently of both.
* An op that is not listed below still raises ``StandInCalled`` (``make_reference_fixtures.py``):
  no number in a fixture can come from an unimplemented stand-in.
* The stand-in itself is checked by the reference's OWN test suite: ``run_reference_tests.py``
  collects ``/root/reference/safe_learning/tests`` on the reference's code behind this module -
  39 of the 44 tests pass (``reference_test_results.json``; the four GP tests run on
  ``numpy_gpflow.py``), the other 5 are refused by name (``tf.gradients``, the Xavier initialiser,
  an optimiser) or skipped (cvxpy).

NumPy-2 / Python-3 compatibility of the reference, applied by ``load_reference``:
``np.int`` -> ``int`` (``functions.py:597``), ``collections.Sequence`` (``lyapunov.py:5``),
``np.column_stack`` / ``np.hstack`` accept generators and ``map`` objects as NumPy 1 did
(``functions.py:635, 1563``, ``utilities.py:147-152``, ``lyapunov.py:50``).
"""

import collections
import collections.abc
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_fixtures as ref_loader          # noqa: E402
sys.path.remove(HERE)

REFERENCE_ROOT = "/root/reference"


# --------------------------------------------------------------------------------------
# Graph nodes
# --------------------------------------------------------------------------------------

def broadcast_shape(a, b):
    if a is None or b is None:
        return None
    out = []
    for x, y in zip(((1,) * (len(b) - len(a)) + tuple(a)), ((1,) * (len(a) - len(b)) + tuple(b))):
        if x is None or y is None:
            out.append(None if 1 in (x, y) or x == y else (x or y))
        else:
            out.append(max(x, y))
    return tuple(out)


def shape_of(x):
    if isinstance(x, Lazy):
        return x.shape
    return np.shape(x)


class Lazy(object):
    """A node of the op graph: ``fn(*args)`` evaluated at ``.eval`` time; a placeholder has no fn."""

    __array_ufunc__ = None          # ndarray (op) Lazy -> Lazy.__r(op)__

    def __init__(self, fn, args=(), shape=None, name=""):
        self.fn, self.args, self.shape, self.name = fn, tuple(args), shape, name

    def eval(self, feed_dict=None, session=None):
        return evaluate(self, feed_dict or {}, {})

    def _binary(self, other, fn, swap=False):
        args = (other, self) if swap else (self, other)
        return Lazy(fn, args, broadcast_shape(shape_of(args[0]), shape_of(args[1])))

    def __add__(self, o): return self._binary(o, np.add)
    def __radd__(self, o): return self._binary(o, np.add, True)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __rsub__(self, o): return self._binary(o, np.subtract, True)
    def __mul__(self, o): return self._binary(o, np.multiply)
    def __rmul__(self, o): return self._binary(o, np.multiply, True)
    def __truediv__(self, o): return self._binary(o, divide)
    def __rtruediv__(self, o): return self._binary(o, divide, True)
    def __neg__(self): return Lazy(np.negative, (self,), self.shape)

    def __getitem__(self, key):
        shape = None
        if self.shape is not None and None not in self.shape:
            shape = np.empty(self.shape)[key].shape
        return Lazy(lambda x: x[key], (self,), shape)


class Variable(Lazy):
    """``tf.Variable``: evaluates to its current array; registered under the enclosing scope."""

    def __init__(self, value, name=None, **_):
        self.value = np.array(value, dtype=np.float64)
        Lazy.__init__(self, lambda: self.value, (), self.value.shape, name or "")
        self.scope = _scope_stack[-1] if _scope_stack else ""
        _variables.append(self)


def divide(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):     # TensorFlow returns inf / nan silently
        return np.true_divide(a, b)


MAP_FN = object()


def evaluate(node, feed_dict, memo):
    if isinstance(node, (list, tuple)):
        return type(node)(evaluate(x, feed_dict, memo) for x in node)
    if not isinstance(node, Lazy):
        return node
    if id(node) in memo:
        return memo[id(node)][1]
    if node.fn is None:
        if node not in feed_dict:
            raise KeyError("placeholder %r was not fed" % node.name)
        value = np.asarray(feed_dict[node])
    elif node.fn is MAP_FN:
        fn, elems = node.args
        rows = evaluate(elems, feed_dict, memo)
        # TensorFlow traces fn once on a symbolic row; tracing it per row builds the same ops
        value = np.array([evaluate(fn(constant(row)), feed_dict, memo) for row in rows])
    else:
        value = node.fn(*[evaluate(a, feed_dict, memo) for a in node.args])
    memo[id(node)] = (node, value)           # holding the node keeps its id from being reused
    return value


def constant(value, dtype=None, **_):
    value = np.asarray(value, dtype=None if dtype is None else dtype.as_numpy_dtype)
    return Lazy(lambda: value, (), value.shape)


def static(shape):
    return None if shape is None else tuple(None if s in (None, -1) else int(s) for s in shape)


# --------------------------------------------------------------------------------------
# Canonical accumulation order
# --------------------------------------------------------------------------------------

def ordered_sum(x, axis=None, keepdims=False):
    """Left-to-right sum along ``axis`` (all axes in C order if None), one rounding per add."""
    x = np.asarray(x)
    if axis is None:
        flat = x.reshape(-1)
        total = flat[0] if len(flat) else x.dtype.type(0)
        for k in range(1, len(flat)):
            total = total + flat[k]
        return np.asarray(total).reshape((1,) * x.ndim) if keepdims else total
    n = x.shape[axis]
    if n == 0:
        return np.sum(x, axis=axis, keepdims=keepdims)
    total = np.take(x, 0, axis=axis)
    for k in range(1, n):
        total = total + np.take(x, k, axis=axis)
    return np.expand_dims(total, axis) if keepdims else total


def ordered_matmul(a, b, transpose_a=False, transpose_b=False):
    """``a @ b`` with every output the left-to-right sum of its products (no FMA)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a = a.T if transpose_a else a
    b = b.T if transpose_b else b
    total = a[:, [0]] * b[[0], :]
    for k in range(1, a.shape[1]):
        total = total + a[:, [k]] * b[[k], :]
    return total


# --------------------------------------------------------------------------------------
# Scopes, variables, sessions
# --------------------------------------------------------------------------------------

_scope_stack = []
_scope_count = collections.Counter()
_variables = []
_named_variables = {}


class Scope(object):
    """``tf.variable_scope`` / ``tf.name_scope``: a fresh name gets a unique
    ``original_name_scope`` ("name_3/"); a string that ends with "/" re-enters that scope
    (``utilities.py:118``: ``tf.variable_scope(self.scope_name)``)."""

    def __init__(self, name, record=True):
        name = getattr(name, "original_name_scope", name)
        if record and not str(name).endswith("/"):
            _scope_count[name] += 1
            name = "%s_%d/" % (name, _scope_count[name])
        self.original_name_scope = str(name)
        self.record = record

    def __enter__(self):
        if self.record:
            _scope_stack.append(self.original_name_scope)
        return self

    def __exit__(self, *exc):
        if self.record:
            _scope_stack.pop()
        return False


def get_collection(key, scope=None):                 # functions.py:58-61
    return [v for v in _variables if scope is None or v.scope.startswith(scope)]


def set_named_variables(arrays):
    """Weights ``tf.get_variable`` hands out by name (LyapunovNetwork, examples/utilities.py:95-99)."""
    _named_variables.clear()
    _named_variables.update({k: Variable(v, name=k) for k, v in arrays.items()})


def get_variable(name, shape=None, dtype=None, initializer=None):
    variable = _named_variables[name]
    assert tuple(shape) == variable.value.shape, (name, shape, variable.value.shape)
    return variable


class Session(object):
    def run(self, fetches, feed_dict=None):
        return evaluate(fetches, feed_dict or {}, {})


def py_func(func, inp, Tout, stateful=True, name=None):          # utilities.py:64
    node = Lazy(lambda *v: func(*v), tuple(inp))
    if not isinstance(Tout, (list, tuple)):
        return node
    if len(Tout) == 1:
        return [Lazy(lambda t: np.asarray(t[0] if isinstance(t, (tuple, list)) else t), (node,))]
    return [Lazy(lambda t, _k=k: np.asarray(t[_k]), (node,)) for k in range(len(Tout))]


def assign(variable, value, name=None, validate_shape=None):
    def store(new):
        variable.value = np.array(new, dtype=np.float64).reshape(variable.value.shape)
        return variable.value
    return Lazy(store, (value,), variable.shape)


def split(value, num_or_size_splits, axis=0):
    shape = shape_of(value)
    if isinstance(num_or_size_splits, int):
        count = num_or_size_splits
        width = None if shape is None or shape[axis] is None else shape[axis] // count
        pieces = []
        for k in range(count):
            piece_shape = None
            if shape is not None:
                piece_shape = tuple(shape[:axis]) + (width,) + tuple(shape[axis + 1:])
            pieces.append(Lazy(lambda v, _k=k: np.split(np.asarray(v), count, axis=axis)[_k],
                               (value,), piece_shape))
        return pieces
    bounds = np.concatenate(([0], np.cumsum(list(num_or_size_splits))))
    pieces = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        key = tuple([slice(None)] * axis + [slice(int(lo), int(hi))])
        piece_shape = None
        if shape is not None:
            piece_shape = tuple(shape[:axis]) + (int(hi - lo),) + tuple(shape[axis + 1:])
        pieces.append(Lazy(lambda v, _key=key: np.asarray(v)[_key], (value,), piece_shape))
    return pieces


def concat(values, axis, name=None):
    values = tuple(values)
    shapes = [shape_of(v) for v in values]
    shape = None
    if all(s is not None for s in shapes):
        shape = list(shapes[0])
        widths = [s[axis] for s in shapes]
        shape[axis] = None if None in widths else sum(widths)
        shape = tuple(shape)
    return Lazy(lambda *v: np.concatenate(v, axis=axis), values, shape)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    sa, sb = shape_of(a), shape_of(b)
    shape = None
    if sa is not None and sb is not None:
        shape = (sa[1] if transpose_a else sa[0], sb[0] if transpose_b else sb[1])
    return Lazy(lambda u, v: ordered_matmul(u, v, transpose_a, transpose_b), (a, b), shape)


def _reduced_shape(shape, axis, keepdims):
    if shape is None:
        return None
    if axis is None:
        return (1,) * len(shape) if keepdims else ()
    axis = axis % len(shape)
    return tuple(shape[:axis]) + ((1,) if keepdims else ()) + tuple(shape[axis + 1:])


def reduce_sum(x, axis=None, keepdims=False, name=None, keep_dims=None):
    keepdims = keepdims if keep_dims is None else keep_dims
    return Lazy(lambda v: ordered_sum(v, axis=axis, keepdims=keepdims), (x,),
                _reduced_shape(shape_of(x), axis, keepdims))


def norm(x, ord=None, axis=None, keepdims=False):                 # lyapunov.py:286
    assert ord == 1
    return Lazy(lambda v: ordered_sum(np.abs(v), axis=axis, keepdims=keepdims), (x,),
                _reduced_shape(shape_of(x), axis, keepdims))


def unary(fn):
    return lambda x, name=None: Lazy(fn, (x,), shape_of(x))


def binary(fn):
    return lambda a, b, name=None: Lazy(fn, (a, b), broadcast_shape(shape_of(a), shape_of(b)))


def tile(x, multiples):
    shape = shape_of(x)
    shape = None if shape is None else tuple(None if s is None else s * m
                                             for s, m in zip(shape, multiples))
    return Lazy(lambda v: np.tile(v, multiples), (x,), shape)


def unstack(x):
    rows = shape_of(x)[0]                    # static, as TensorFlow requires as well
    return [x[k] for k in range(rows)]


def install(tf):
    """Replace the raising placeholders of the ``tensorflow`` stand-in module by the ops above."""
    dtype = lambda np_type: types.SimpleNamespace(as_numpy_dtype=np_type)   # noqa: E731
    tf.float64, tf.int32, tf.int64, tf.bool = (dtype(np.float64), dtype(np.int32), dtype(np.int64),
                                               dtype(np.bool_))
    tf.Tensor, tf.Variable = Lazy, Variable                          # utilities.py:137
    tf.GraphKeys = types.SimpleNamespace(TRAINABLE_VARIABLES="trainable_variables")
    tf.get_collection = get_collection
    tf.get_variable = get_variable
    tf.variables_initializer = lambda variables: Lazy(lambda: None, ())
    tf.variable_scope = lambda name, **kw: Scope(name)
    tf.name_scope = lambda name: Scope(name, record=False)           # utilities.py:108
    tf.make_template = lambda name, func, **kw: func                 # functions.py:49
    tf.get_default_session = lambda: Session()
    tf.py_func = py_func
    tf.assign = assign
    tf.stop_gradient = lambda x: x
    tf.placeholder = lambda dt, shape=None, name="": Lazy(None, (), static(shape), name)
    tf.constant = constant
    tf.eye = lambda n, dtype=None: Lazy(lambda k: np.eye(int(k)), (n,))
    tf.shape = lambda x: Lazy(lambda v: np.array(np.shape(v)), (x,),
                              None if shape_of(x) is None else (len(shape_of(x)),))
    # functions.py:408-409, 441: LAPACK (dpotrf / dtrtrs) for TensorFlow's Eigen LLT / triangular
    # solve - both backward-stable factorizations, results agree up to rounding x cond(K)
    tf.cholesky = lambda a, name=None: Lazy(lambda v: scipy.linalg.cholesky(v, lower=True), (a,),
                                            shape_of(a))
    tf.matrix_triangular_solve = lambda a, b, lower=True, name=None: Lazy(
        lambda u, v: scipy.linalg.solve_triangular(u, v, lower=lower), (a, b), shape_of(b))
    tf.sqrt, tf.exp = unary(np.sqrt), unary(np.exp)
    tf.expand_dims = lambda x, axis: Lazy(lambda v: np.expand_dims(v, axis), (x,))
    tf.fill = lambda dims, value: Lazy(lambda d, v: np.full(tuple(int(k) for k in d), v), (dims, value))
    tf.zeros = lambda dims, dtype=None: Lazy(lambda d: np.zeros(tuple(int(k) for k in d)), (dims,))
    tf.matmul, tf.reduce_sum, tf.norm = matmul, reduce_sum, norm
    tf.split, tf.concat, tf.tile, tf.unstack = split, concat, tile, unstack
    tf.stack = lambda xs, axis=0, name=None: (
        constant(np.stack(xs, axis=axis)) if isinstance(xs, np.ndarray)
        else Lazy(lambda *v: np.stack(v, axis=axis), tuple(xs)))
    tf.gather = lambda params, indices, validate_indices=None: Lazy(
        lambda p, i: np.asarray(p)[np.asarray(i)], (params, indices))
    tf.less, tf.maximum, tf.minimum = binary(np.less), binary(np.maximum), binary(np.minimum)
    tf.sin, tf.cos, tf.tanh, tf.abs = unary(np.sin), unary(np.cos), unary(np.tanh), unary(np.abs)
    tf.square, tf.ceil, tf.is_nan = unary(np.square), unary(np.ceil), unary(np.isnan)
    tf.zeros_like = unary(np.zeros_like)
    tf.squeeze = lambda x, axis=None: Lazy(lambda v: np.squeeze(v, axis=axis), (x,))
    tf.reduce_min = lambda x: Lazy(np.min, (x,), ())
    tf.reduce_all = lambda x: Lazy(np.all, (x,), ())
    tf.where = lambda c, a, b: Lazy(np.where, (c, a, b), shape_of(a))
    tf.cast = lambda x, dt: Lazy(lambda v: np.asarray(v).astype(dt.as_numpy_dtype), (x,), shape_of(x))
    tf.reshape = lambda x, shape: Lazy(lambda v: np.reshape(v, shape), (x,), static(shape))
    plain_tile = tf.tile
    tf.tile = lambda x, multiples: (plain_tile(x, multiples) if not any(isinstance(m, Lazy) for m in multiples)
                                    else Lazy(lambda v, *m: np.tile(v, [int(k) for k in m]),
                                              (x,) + tuple(multiples)))
    tf.linspace = lambda a, b, n: Lazy(lambda u, v, k: np.linspace(u, v, int(k)), (a, b, n), (None,))
    tf.meshgrid = lambda *xs, **kw: [Lazy(lambda *v, _k=k: np.meshgrid(*v, **kw)[_k], tuple(xs))
                                     for k in range(len(xs))]
    tf.map_fn = lambda fn, elems, dtype=None, parallel_iterations=None: Lazy(MAP_FN, (fn, elems))


def lazy_function(fn, ncols, uncertain=False):
    """A NumPy callable as a graph function: Lazy inputs -> Lazy output(s) with ``ncols`` columns
    (``(mean, error)`` for an uncertain dynamics model, ``lyapunov.py:340``)."""
    def call(*inputs):
        if not uncertain:
            return Lazy(lambda *v: np.asarray(fn(*v)), inputs, (None, ncols))
        pair = Lazy(lambda *v: fn(*v), inputs)
        return (Lazy(lambda t: t[0], (pair,), (None, ncols)),
                Lazy(lambda t: t[1], (pair,), (None, ncols)))
    return call


# --------------------------------------------------------------------------------------
# Loading the reference's modules
# --------------------------------------------------------------------------------------

def _load(module_name, path):
    spec = importlib.util.spec_from_file_location(module_name, path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[module_name] = module
    ref_loader._armed[0] = False             # decorators / default arguments may touch stand-ins
    try:
        spec.loader.exec_module(module)
    finally:
        ref_loader._armed[0] = True
    return module


def load_reference(examples=False, gpflow=True):
    """-> namespace(functions, lyapunov, reinforcement_learning, examples, config): the
    reference's modules, executed from ``/root/reference`` with this stand-in as ``tensorflow`` and
    (``gpflow=True``) ``numpy_gpflow`` as ``gpflow``: the reference's ``GPRCached`` /
    ``GaussianProcess`` are then usable (``functions.py:357-546``)."""
    collections.Sequence = collections.abc.Sequence                   # lyapunov.py:5
    for name in ("column_stack", "hstack"):
        plain = getattr(np, name)
        if not getattr(plain, "_accepts_iterators", False):
            wrapped = (lambda f: lambda tup: f(tup if isinstance(tup, np.ndarray) else tuple(tup)))(plain)
            wrapped._accepts_iterators = True
            setattr(np, name, wrapped)
    gpflow_module = None
    if gpflow:
        sys.path.insert(0, HERE)
        import numpy_gpflow
        sys.path.remove(HERE)
        gpflow_module = numpy_gpflow.module()
    functions = ref_loader.load_reference(gpflow_module)
    tf = sys.modules["tensorflow"]
    install(tf)
    package = sys.modules["safe_learning"]
    ref = os.path.join(REFERENCE_ROOT, "safe_learning")
    out = types.SimpleNamespace(functions=functions, config=package.config, examples=None)
    out.lyapunov = _load("safe_learning.lyapunov", os.path.join(ref, "lyapunov.py"))
    out.reinforcement_learning = _load("safe_learning.reinforcement_learning",
                                       os.path.join(ref, "reinforcement_learning.py"))
    if examples:
        for name in ("DeterministicFunction", "GridWorld"):           # examples/utilities.py:12
            setattr(package, name, getattr(functions, name))
        out.examples = _load("reference_examples_utilities",
                             os.path.join(REFERENCE_ROOT, "examples", "utilities.py"))
    return out


def install_test_extras(tf):
    """What the reference's own TESTS use on top of the hot path (run_reference_tests.py)."""
    default = tf.get_default_graph()
    tf.Graph = lambda: types.SimpleNamespace()                      # a holder for a feed dict
    tf.Session = lambda graph=None, **kwargs: _SessionContext(graph or default)
    tf.get_default_session = lambda: _SessionContext(default)
    tf.reset_default_graph = lambda: None
    tf.float32 = types.SimpleNamespace(as_numpy_dtype=np.float32)
    placeholder = tf.placeholder
    tf.placeholder = lambda dtype=None, shape=None, name="": placeholder(dtype, shape, name)
    tf.global_variables_initializer = lambda: Lazy(lambda: None, ())


class _SessionContext(Session):
    def __init__(self, graph):
        self.graph = graph

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def as_default(self):
        return self
