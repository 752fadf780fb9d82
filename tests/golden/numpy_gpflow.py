"""A stand-in for the part of gpflow==0.4.0 the reference's GP classes sit on (build container only).

``/root/reference/safe_learning/functions.py:357-546`` defines ``GPRCached(gpflow.gpr.GPR)`` and
``GaussianProcess``; gpflow (pinned ``gpflow==0.4.0`` by the reference's ``requirements.txt:3`` and
``setup.py:41``) is a third-party dependency that is absent from ``/root/reference`` and from this
image.  With this module as ``gpflow`` - installed BEFORE ``functions.py`` is executed, so that
``GPRCached`` really inherits from the ``GPR`` below - the reference's own
``GPRCached._compute_cache`` (scale, Cholesky, triangular solve, ``:395-410``), ``build_predict``
(``:417-458``: scaled kernel, solve, mean, variance, ``/scale``, tile),
``GaussianProcess.build_evaluation`` (``:507-515``), ``add_data_point`` (``:525-546``) and
``FunctionStack`` (``:254-307``) run UNMODIFIED from the checkout; every op they request from
TensorFlow is answered by ``numpy_tf`` (``tf.cholesky`` / ``tf.matrix_triangular_solve`` -> LAPACK
through SciPy).

What is restated here, from gpflow 0.4.0's published sources, and nothing more:

* ``kernels.RBF`` / ``Matern32`` / ``Linear`` / ``Add`` / ``Prod`` with ``active_dims``
  (``Stationary.square_dist``: ``-2 (X/l)(X2/l)^T + |X/l|^2 + |X2/l|^2``;
  ``K = variance * exp(-square_dist / 2)``; ``Kdiag = fill(variance)``);
* ``mean_functions.Zero`` (``zeros([N, 1])``) and ``Linear`` (``X A + b``);
* ``likelihoods.Gaussian`` (a ``variance`` parameter, default 1.0);
* ``gpr.GPR`` (``__init__``: data holders, likelihood, ``num_latent``; ``build_predict``: the
  uncached posterior - Cholesky of ``K + variance I``, two triangular solves, mean, marginal variance);
* the parameter plumbing the reference touches: ``param.DataHolder`` (``.value``, ``.shape``,
  assignment of an array to the attribute replaces the data), ``param.Param``, ``param.AutoFlow``
  (runs a method's graph on NumPy arguments), ``tf_mode``, ``make_tf_array``, ``get_free_state``,
  ``get_feed_dict_keys``, ``update_feed_dict``, ``predict_f``.

The restatement is pinned by the reference's OWN tests, which ``run_reference_tests.py`` now runs on
it: ``test_functions.py:237-261`` (known posterior mean / 3-sigma bound of a plain ``GPR`` after
``add_data_point``), ``:164-200`` (the reference's ``GPRCached`` against that ``GPR``, with and
without added data) and ``:220-235``.
"""

import contextlib
import sys
import types

import numpy as np

import numpy_tf
from numpy_tf import Lazy


def _tf():
    return sys.modules["tensorflow"]


class _Holder(Lazy):
    """A node that evaluates to the array it currently holds (shape follows the data)."""

    def __init__(self, value):
        self._value = np.array(value, dtype=np.float64)
        Lazy.__init__(self, lambda: self._value, ())

    @property
    def shape(self):
        return self._value.shape

    @shape.setter
    def shape(self, _):
        pass

    @property
    def value(self):
        return self._value

    def set_data(self, value):
        self._value = np.array(value, dtype=np.float64)


class DataHolder(_Holder):
    """``gpflow.param.DataHolder``: data that is fed, not optimised."""

    def __init__(self, array, on_shape_change="raise"):
        _Holder.__init__(self, array)


class Param(_Holder):
    """``gpflow.param.Param``: a (free) parameter; transforms are irrelevant without optimisation."""

    def __init__(self, value, transform=None):
        _Holder.__init__(self, value)


class Parameterized(object):
    """Assigning an array / number to an attribute that holds a Param or DataHolder replaces its
    value (gpflow's ``Parameterized.__setattr__``: ``gp.X = ...``, ``gp.likelihood.variance = ...``)."""

    def __setattr__(self, key, value):
        current = self.__dict__.get(key)
        if isinstance(current, _Holder) and not isinstance(value, _Holder):
            current.set_data(value)
        else:
            object.__setattr__(self, key, value)

    def _params(self):
        for key in sorted(self.__dict__):
            item = self.__dict__[key]
            if isinstance(item, Param):
                yield item
            elif isinstance(item, Parameterized):
                for sub in item._params():
                    yield sub

    # ---- what GaussianProcess.__init__ / update_feed_dict call (functions.py:492-523) ----
    def make_tf_array(self, free_array):
        return 0

    def get_free_state(self):
        return np.concatenate([np.ravel(p.value) for p in self._params()] or [np.zeros(0)])

    def get_feed_dict_keys(self):
        return {}

    def update_feed_dict(self, keys, feed_dict):
        pass

    @contextlib.contextmanager
    def tf_mode(self):
        yield


def AutoFlow(*tf_arg_tuples):
    """``gpflow.param.AutoFlow``: the decorated method builds a graph from placeholders of the
    given types; calling it with NumPy arrays runs that graph."""
    def decorate(method):
        def runner(instance, *np_args):
            nodes = method(instance, *[numpy_tf.constant(np.asarray(a, dtype=np.float64))
                                       for a in np_args])
            return numpy_tf.evaluate(nodes, {}, {})
        runner.__name__ = getattr(method, "__name__", "autoflow")
        return runner
    return decorate


# ---- mean functions (gpflow/mean_functions.py) -------------------------------------------------

class Zero(Parameterized):
    def __call__(self, X):
        tf = _tf()
        return tf.zeros(tf.stack([tf.shape(X)[0], 1]), dtype=tf.float64)


class Linear(Parameterized):
    def __init__(self, A=None, b=None):
        self.A = Param(np.ones((1, 1)) if A is None else np.atleast_2d(A))
        self.b = Param(np.zeros(1) if b is None else np.atleast_1d(b))

    def __call__(self, X):
        return _tf().matmul(X, self.A) + self.b


# ---- kernels (gpflow/kernels.py: Kern, Stationary, RBF, Matern32, Linear, Add, Prod) -------------

class Kern(Parameterized):
    """``Kern.__init__`` / ``_slice`` / ``__add__`` / ``__mul__`` of gpflow 0.4.0: a kernel reads the
    columns ``active_dims`` of its inputs (default ``slice(input_dim)``)."""

    def _init_dims(self, input_dim, active_dims):
        self.input_dim = int(input_dim)
        object.__setattr__(self, "active_dims", slice(self.input_dim) if active_dims is None
                           else np.asarray(list(active_dims), dtype=np.int64))

    def _slice(self, X, X2):
        X = X[:, self.active_dims]
        return X, (None if X2 is None else X2[:, self.active_dims])

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])


class Stationary(Kern):
    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        self._init_dims(input_dim, active_dims)
        self.variance = Param(variance)
        if ARD:
            lengthscales = np.ones(self.input_dim) if lengthscales is None else \
                np.asarray(lengthscales, dtype=np.float64) * np.ones(self.input_dim)
        elif lengthscales is None:
            lengthscales = 1.0
        self.lengthscales = Param(lengthscales)
        self.ARD = ARD

    def square_dist(self, X, X2):
        tf = _tf()
        X = X / self.lengthscales
        Xs = tf.reduce_sum(tf.square(X), 1)
        if X2 is None:
            return -2 * tf.matmul(X, X, transpose_b=True) + \
                tf.reshape(Xs, (-1, 1)) + tf.reshape(Xs, (1, -1))
        X2 = X2 / self.lengthscales
        X2s = tf.reduce_sum(tf.square(X2), 1)
        return -2 * tf.matmul(X, X2, transpose_b=True) + \
            tf.reshape(Xs, (-1, 1)) + tf.reshape(X2s, (1, -1))

    def euclid_dist(self, X, X2):
        return _tf().sqrt(self.square_dist(X, X2) + 1e-12)

    def Kdiag(self, X):
        tf = _tf()
        return tf.fill(tf.stack([tf.shape(X)[0]]), tf.squeeze(self.variance))


class RBF(Stationary):
    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        return self.variance * _tf().exp(-self.square_dist(X, X2) / 2)


class Matern32(Stationary):
    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        r = self.euclid_dist(X, X2)
        return self.variance * (1. + np.sqrt(3.) * r) * _tf().exp(-np.sqrt(3.) * r)


class LinearKernel(Kern):
    """``kernels.Linear`` (the mean function of the same name is ``Linear`` above)."""

    def __init__(self, input_dim, variance=1.0, active_dims=None, ARD=False):
        self._init_dims(input_dim, active_dims)
        self.ARD = ARD
        self.variance = Param(np.ones(self.input_dim) * variance if ARD else variance)

    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        return _tf().matmul(X * self.variance, X if X2 is None else X2, transpose_b=True)

    def Kdiag(self, X):
        tf = _tf()
        X, _ = self._slice(X, None)
        return tf.reduce_sum(tf.square(X) * self.variance, 1)


class Add(Kern):
    def __init__(self, kern_list):
        kerns = []
        for k in kern_list:                              # Combination: members of the same kind flatten
            kerns.extend(k.kern_list if type(k) is type(self) else [k])
        object.__setattr__(self, "kern_list", kerns)
        for i, k in enumerate(kerns):
            setattr(self, "kern_%d" % i, k)

    def K(self, X, X2=None):
        out = self.kern_list[0].K(X, X2)
        for k in self.kern_list[1:]:
            out = out + k.K(X, X2)
        return out

    def Kdiag(self, X):
        out = self.kern_list[0].Kdiag(X)
        for k in self.kern_list[1:]:
            out = out + k.Kdiag(X)
        return out


class Prod(Add):
    def K(self, X, X2=None):
        out = self.kern_list[0].K(X, X2)
        for k in self.kern_list[1:]:
            out = out * k.K(X, X2)
        return out

    def Kdiag(self, X):
        out = self.kern_list[0].Kdiag(X)
        for k in self.kern_list[1:]:
            out = out * k.Kdiag(X)
        return out


class Gaussian(Parameterized):
    def __init__(self):
        self.variance = Param(1.0)


# ---- gpflow/model.py::GPModel, gpflow/gpr.py::GPR ----------------------------------------------

class GPR(Parameterized):
    def __init__(self, X, Y, kern, mean_function=None, name="name"):
        self.name = name
        self.kern = kern
        self.likelihood = Gaussian()
        self.mean_function = mean_function if mean_function is not None else Zero()
        object.__setattr__(self, "X", DataHolder(X, on_shape_change="pass"))
        object.__setattr__(self, "Y", DataHolder(Y, on_shape_change="pass"))
        self.num_latent = self.Y.shape[1]

    def build_predict(self, Xnew, full_cov=False):
        tf = _tf()
        assert not full_cov
        Kx = self.kern.K(self.X, Xnew)
        K = self.kern.K(self.X) + tf.eye(tf.shape(self.X)[0], dtype=tf.float64) * self.likelihood.variance
        L = tf.cholesky(K)
        A = tf.matrix_triangular_solve(L, Kx, lower=True)
        V = tf.matrix_triangular_solve(L, self.Y - self.mean_function(self.X))
        fmean = tf.matmul(A, V, transpose_a=True) + self.mean_function(Xnew)
        fvar = self.kern.Kdiag(Xnew) - tf.reduce_sum(tf.square(A), 0)
        fvar = tf.tile(tf.reshape(fvar, (-1, 1)), [1, tf.shape(self.Y)[1]])
        return fmean, fvar

    @AutoFlow((np.float64, [None, None]))
    def predict_f(self, Xnew):
        return self.build_predict(Xnew)


def module():
    """-> the stand-in as a ``gpflow`` module tree (what ``functions.py`` and the reference's tests
    import: ``gpflow.gpr.GPR``, ``gpflow.param.*``, ``gpflow.kernels.RBF``, ``gpflow.mean_functions``)."""
    gpflow = types.ModuleType("gpflow")
    gpflow.gpr = types.ModuleType("gpflow.gpr")
    gpflow.gpr.GPR = GPR
    gpflow.param = types.ModuleType("gpflow.param")
    gpflow.param.DataHolder, gpflow.param.Param = DataHolder, Param
    gpflow.param.AutoFlow, gpflow.param.Parameterized = AutoFlow, Parameterized
    gpflow.kernels = types.ModuleType("gpflow.kernels")
    gpflow.kernels.RBF, gpflow.kernels.Matern32, gpflow.kernels.Linear = RBF, Matern32, LinearKernel
    gpflow.kernels.Add, gpflow.kernels.Prod = Add, Prod
    gpflow.mean_functions = types.ModuleType("gpflow.mean_functions")
    gpflow.mean_functions.Zero, gpflow.mean_functions.Linear = Zero, Linear
    gpflow.likelihoods = types.ModuleType("gpflow.likelihoods")
    gpflow.likelihoods.Gaussian = Gaussian
    return gpflow
