"""Function objects of the Lyapunov sweep as declarative specs for the HIP engine.

The reference composes TensorFlow callables (``safe_learning/functions.py``); a GPU kernel
cannot call back into Python, so here every class is a small parameter record with the
reference's constructor signature that knows how to write itself into the C-ABI model
description (``include/sl_hip.h``).  All arithmetic happens in the HIP kernels; this module
only prepares host-side metadata (grids, Cholesky factors, unit-cell triangulations), which
is also host-side NumPy/SciPy work in the reference.

Reference classes mirrored (constructor arguments keep their names):
``GridWorld`` (functions.py:579-817), ``QuadraticFunction`` (:1513-1543), ``LinearSystem``
(:1546-1583), ``Saturation`` (:310-354), ``GPRCached`` (:357-458), ``GaussianProcess``
(:461-546), ``FunctionStack`` (:254-307), ``Triangulation`` (:981-1510) and, from
``examples/utilities.py``, ``InvertedPendulum`` (:144-289), ``CartPole`` (:292-437),
``LyapunovNetwork`` (:48-104).
"""

from itertools import product as _cartesian

import numpy as np
import scipy.linalg
import scipy.signal
from scipy import spatial

import itertools

from . import _hip
from .configuration import config
from .unit_cells import UNIT_CELL_SIMPLICES


def qhull_unit_cell(unit_maxes):
    """The unit-cell triangulation as the reference computes it (``functions.py:1019-1023``):
    ``spatial.Delaunay`` of ``cartesian(*np.diag(unit_maxes))`` (3^d rows with duplicates for
    d >= 3), as corner codes (bit k = coordinate k at ``unit_maxes[k]``)."""
    unit_maxes = np.asarray(unit_maxes, dtype=config.np_dtype)
    d = len(unit_maxes)
    corners = np.array(list(_cartesian(*np.diag(unit_maxes))), dtype=config.np_dtype)
    tri = spatial.Delaunay(corners)
    codes = ((corners > 0).astype(np.int64) << np.arange(d)).sum(axis=1)
    return codes[tri.simplices].astype(np.int32)

# Upload caches (``_model.ModelBuilder``) are keyed on these process-wide, never reused tokens:
# ``id()`` values can be recycled by CPython once an object is collected.
_TOKENS = itertools.count(1)

__all__ = ['GridWorld', 'DimensionError', 'DeterministicFunction', 'UncertainFunction',
           'QuadraticFunction', 'LinearSystem', 'Saturation', 'AbsFunction', 'Norm1Function',
           'AbsGradient', 'Gradient', 'ConstantFunction', 'RBF', 'Matern32', 'Linear', 'GPRCached', 'GaussianProcess',
           'FunctionStack', 'Triangulation', 'InvertedPendulum', 'CartPole', 'LyapunovNetwork',
           'NeuralNetwork']


class DimensionError(Exception):
    """Same role as ``safe_learning/functions.py:575-576``."""


# ----------------------------------------------------------------------------------------------
# grid
# ----------------------------------------------------------------------------------------------

class GridWorld(object):
    """Regular grid over a hyper-rectangle (reference: ``functions.py:591-620``).

    ``limits`` is ``[(lo, hi), ...]``, ``num_points`` an int or one int per dimension.  Flat
    indices are C-order (last dimension fastest); state ``k`` of index ``ijk`` is
    ``ijk[k] * unit_maxes[k] + offset[k]`` - the addressing contract of the HIP kernels.
    """

    def __init__(self, limits, num_points):
        dtype = config.np_dtype
        self.limits = np.atleast_2d(limits).astype(dtype)
        self.ndim = len(self.limits)
        self.num_points = np.broadcast_to(num_points, self.ndim).astype(np.int64)
        if np.any(self.num_points < 2):
            raise DimensionError('There must be at least 2 points in each dimension.')
        lower, upper = self.limits[:, 0], self.limits[:, 1]
        self.offset = lower
        self.unit_maxes = ((upper - lower) / (self.num_points - 1)).astype(dtype)
        self.offset_limits = np.stack((np.zeros_like(lower), upper - lower), axis=1)
        self.discrete_points = [np.linspace(lo, hi, n, dtype=dtype)
                                for (lo, hi), n in zip(self.limits, self.num_points)]
        self.nrectangles = int(np.prod(self.num_points - 1))
        self.nindex = int(np.prod(self.num_points))
        self._all_points = None

    def __len__(self):
        return self.nindex

    @property
    def all_points(self):
        """``[nindex, ndim]`` array of every grid point (``functions.py:622-638``).

        Only for small grids / plotting: the kernels generate states from indices instead."""
        if self._all_points is None:
            mesh = np.meshgrid(*self.discrete_points, indexing='ij')
            self._all_points = np.stack([m.ravel() for m in mesh], axis=1).astype(config.np_dtype)
        return self._all_points

    def sample_continuous(self, num_samples):
        """Uniform samples from the continuous domain (``functions.py:644-659``)."""
        rand = np.random.uniform(0, 1, size=(num_samples, self.ndim))
        return rand * np.diff(self.limits, axis=1).T + self.offset

    def sample_discrete(self, num_samples, replace=False):
        """Uniform samples from the grid points (``functions.py:661-677``)."""
        idx = np.random.choice(self.nindex, size=num_samples, replace=replace)
        return self.index_to_state(idx)

    def _check_dimensions(self, states):
        if not states.shape[1] == self.ndim:
            raise DimensionError('the input argument has the wrong dimensions.')

    def _center_states(self, states, clip=True):
        """States relative to the lower corner, optionally clipped 2 eps inside
        (``functions.py:691-712``)."""
        eps = np.finfo(config.np_dtype).eps
        states = np.atleast_2d(states).astype(config.np_dtype) - self.offset[None, :]
        if clip:
            np.clip(states, self.offset_limits[:, 0] + 2 * eps,
                    self.offset_limits[:, 1] - 2 * eps, out=states)
        return states

    def index_to_state(self, indices):
        """Flat indices -> states (``functions.py:714-731``)."""
        ijk = np.stack(np.unravel_index(np.atleast_1d(indices), self.num_points), axis=1)
        return ijk.astype(config.np_dtype) * self.unit_maxes + self.offset

    def state_to_index(self, states):
        """Nearest grid point of each state, clipped to the domain (``functions.py:733-752``)."""
        states = np.atleast_2d(states)
        self._check_dimensions(states)
        clipped = np.clip(states, self.limits[:, 0], self.limits[:, 1])
        ijk = np.rint((clipped - self.offset) * (1. / self.unit_maxes)).astype(np.int32)
        return np.ravel_multi_index(ijk.T, self.num_points)

    def state_to_rectangle(self, states):
        """Index of the grid cell containing each state (``functions.py:754-776``)."""
        cells = []
        for k in range(self.ndim):
            idx = np.digitize(states[:, k], self.discrete_points[k]) - 1
            cells.append(np.clip(idx, 0, self.num_points[k] - 2))
        return np.ravel_multi_index(cells, self.num_points - 1)

    def rectangle_to_state(self, rectangles):
        """Lower-left corner state of each cell (``functions.py:778-798``)."""
        ijk = np.stack(np.unravel_index(np.atleast_1d(rectangles), self.num_points - 1), axis=1)
        return ijk.astype(config.np_dtype) * self.unit_maxes + self.offset

    def rectangle_corner_index(self, rectangles):
        """Flat grid index of each cell's lower-left corner (``functions.py:800-817``)."""
        ijk = np.unravel_index(rectangles, self.num_points - 1)
        return np.ravel_multi_index(np.atleast_2d(ijk), self.num_points)

    # ---- C-ABI view ------------------------------------------------------------------------
    def _desc(self):
        if self.ndim > _hip.MAX_STATE_DIM:
            raise DimensionError('the HIP engine supports at most %d state dimensions'
                                 % _hip.MAX_STATE_DIM)
        g = _hip.GridDesc()
        g.d = self.ndim
        for k in range(self.ndim):
            g.num_points[k] = int(self.num_points[k])
            g.offset[k] = float(self.offset[k])
            g.unit_maxes[k] = float(self.unit_maxes[k])
            g.upper[k] = float(self.limits[k, 1])
        return g


# ----------------------------------------------------------------------------------------------
# spec base classes
# ----------------------------------------------------------------------------------------------

class Function(object):
    """Base of all specs.  ``-f`` is supported as a sign flag (``functions.py:120-122``)."""

    negate = False

    def __neg__(self):
        import copy
        other = copy.copy(self)
        other.negate = not self.negate
        return other

    _role = None      # 'value', 'policy', 'dynamics' or 'linear'

    def __call__(self, *points):
        """Evaluate at explicit points on the GPU (``functions.py:63-82``); several inputs are
        concatenated like ``concatenate_inputs`` does (``utilities.py:123-159``)."""
        from . import _evaluate
        arrays = [np.atleast_2d(np.asarray(p, dtype=config.np_dtype)) for p in points]
        if self._role == 'value':
            out = _evaluate.value(self, np.hstack(arrays))
            return out
        if self._role == 'policy':
            return _evaluate.policy(self, np.hstack(arrays))
        if self._role == 'dynamics':
            joined = np.hstack(arrays)
            d = self.output_dim
            return _evaluate.dynamics(self, joined[:, :d], joined[:, d:])
        if self._role == 'linear':
            joined = np.hstack(arrays)
            if len(arrays) == 2 and self.matrix.shape[0] == arrays[0].shape[1]:
                return _evaluate.dynamics(self, arrays[0], arrays[1])
            if self.matrix.shape[0] <= _hip.MAX_ACTION_DIM and len(arrays) == 1:
                return _evaluate.policy(self, joined)
            return _evaluate.linear_map(self, joined)
        raise NotImplementedError('%s cannot be evaluated on its own' % type(self).__name__)


class DeterministicFunction(Function):
    """Marker base class (``functions.py:233-238``)."""


class UncertainFunction(Function):
    """Marker base class: evaluation yields ``(mean, error_bound)`` (``functions.py:202-230``)."""


def _hstack_matrices(matrices):
    if isinstance(matrices, np.ndarray):
        matrices = (matrices,)
    return np.hstack([np.atleast_2d(m).astype(config.np_dtype) for m in matrices])


class QuadraticFunction(DeterministicFunction):
    """``x P x^T`` (``functions.py:1513-1543``); ``P`` need not be symmetric."""

    _role = 'value'

    def __init__(self, matrix, name='quadratic'):
        self.matrix = np.atleast_2d(matrix).astype(config.np_dtype)
        self.ndim = self.matrix.shape[0]
        self.name = name

    def gradient_function(self):
        """``LinearSystem`` computing ``x (P + P^T)`` (``functions.py:1541-1543``)."""
        return LinearSystem((self.matrix + self.matrix.T,))

    def _write_value(self, desc):
        n = self.ndim
        if n > _hip.MAX_INPUT_DIM:
            raise DimensionError('quadratic form too large for the HIP engine')
        desc.kind = _hip.V_QUADRATIC
        desc.negate = int(self.negate)
        for i in range(n):
            for j in range(n):
                desc.matrix[i][j] = float(self.matrix[i, j])


class LinearSystem(DeterministicFunction):
    """``[x, u] M^T`` with ``M = hstack(matrices)`` (``functions.py:1546-1583``)."""

    _role = 'linear'

    def __init__(self, matrices, name='linear_system'):
        self.matrix = _hstack_matrices(matrices)
        self.output_dim, self.input_dim = self.matrix.shape
        self.name = name


class Saturation(DeterministicFunction):
    """Clamp ``fun`` to ``[lower, upper]`` (``functions.py:310-354``)."""

    _role = 'policy'

    def __init__(self, fun, lower, upper, name='saturation'):
        self.fun, self.lower, self.upper = fun, lower, upper
        self.input_dim = getattr(fun, 'input_dim', None)
        self.output_dim = getattr(fun, 'output_dim', None)
        self.name = name


class ConstantFunction(DeterministicFunction):
    """The same output row for every input (``functions.py:241-251``)."""

    def __init__(self, constant, name='constant_function'):
        self.constant = np.atleast_1d(np.asarray(constant, dtype=config.np_dtype))
        self.output_dim = len(self.constant)
        self.name = name


class AbsFunction(DeterministicFunction):
    """``|fun(x)|`` - replaces ``lambda x: tf.abs(grad(x))`` of the notebooks."""

    def __init__(self, fun):
        self.fun = fun


class Norm1Function(DeterministicFunction):
    """``||fun(x)||_1`` - replaces ``lambda x: tf.norm(grad(x), ord=1, axis=1, keepdims=True)``.

    ``constant + Norm1Function(LinearSystem(M))`` gives the affine form ``c + ||M x||_1`` that the
    engine accepts as a state-dependent ``lipschitz_dynamics`` (``lyapunov.py:227-244``)."""

    def __init__(self, fun, constant=0.0):
        self.fun = fun
        self.constant = float(constant)

    def __add__(self, constant):
        if not np.isscalar(constant):
            return NotImplemented
        return Norm1Function(self.fun, self.constant + float(constant))

    __radd__ = __add__


class Gradient(DeterministicFunction):
    """``dV/dx`` of a table / network value function: ``tri.gradient`` (``functions.py:1302-1326``)
    or ``tf.gradients(V(x), x)`` of the notebooks.  Use inside ``AbsFunction`` / ``Norm1Function``
    as the local Lipschitz constant ``L_v``."""

    def __init__(self, fun):
        self.fun = fun


def AbsGradient(fun):
    """``|dV/dx|`` per dimension (``inverted_pendulum.ipynb`` cell 14)."""
    return AbsFunction(Gradient(fun))


# ----------------------------------------------------------------------------------------------
# Gaussian process
# ----------------------------------------------------------------------------------------------

class Kern(object):
    """Kernels with the gpflow 0.4.0 ``kernels.py`` surface the reference relies on: ``K(X, X2)``,
    ``Kdiag(X)`` (``functions.py:401, 438, 445, 450``), ``active_dims``, ``+`` and ``*``
    (``examples/inverted_pendulum.ipynb:152-158``: ``Linear + Matern32 * Linear``).  The Gram
    matrix of the training set is formed here on the host; the engine evaluates ``k(X, x)`` and
    ``k(x, x)`` of the grid cells from :meth:`_factors`."""

    def __init__(self, input_dim, active_dims=None):
        self.input_dim = int(input_dim)
        if active_dims is None:
            active_dims = range(self.input_dim)          # gpflow: slice(input_dim)
        self.active_dims = np.asarray(list(active_dims), dtype=np.int64)
        if len(self.active_dims) != self.input_dim:
            raise ValueError('active_dims has %d entries, input_dim is %d'
                             % (len(self.active_dims), self.input_dim))

    def _slice(self, X, X2):
        X = np.asarray(X, dtype=config.np_dtype)[:, self.active_dims]
        X2 = X if X2 is None else np.asarray(X2, dtype=config.np_dtype)[:, self.active_dims]
        return X, X2

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])

    def _products(self):
        """The kernel as a sum of products of leaves: ``[[leaf, ...], ...]``."""
        return [[self]]

    def _factors(self, p):
        """``[(kind, product, variance[p], inv_lengthscales[p])]`` for ``sl_gp_set_head_kernel``."""
        out = []
        for number, product in enumerate(self._products()):
            for leaf in product:
                variance, inv_ls = np.zeros(p), np.zeros(p)
                if leaf.active_dims.max() >= p:
                    raise ValueError('kernel reads input column %d of %d' % (leaf.active_dims.max(), p))
                leaf._fill(variance, inv_ls)
                out.append((leaf._kind, number, variance, inv_ls))
        return out


class _Stationary(Kern):
    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        Kern.__init__(self, input_dim, active_dims)
        self.variance = float(variance)
        if lengthscales is None:
            lengthscales = 1.0
        self.lengthscales = np.broadcast_to(np.asarray(lengthscales, dtype=config.np_dtype),
                                            (self.input_dim,)).copy()
        self.ARD = bool(ARD)

    def _square_dist(self, X, X2):
        X, X2 = self._slice(X, X2)
        diff = X[:, None, :] / self.lengthscales - X2[None, :, :] / self.lengthscales
        return np.einsum('ijk,ijk->ij', diff, diff)

    def Kdiag(self, X):
        return np.full(len(X), self.variance, dtype=config.np_dtype)

    def _fill(self, variance, inv_ls):
        variance[0] = self.variance
        inv_ls[self.active_dims] = 1.0 / self.lengthscales


class RBF(_Stationary):
    """Squared-exponential kernel with the gpflow 0.4.0 ``kernels.RBF`` signature:
    ``k(x, x') = variance * exp(-0.5 * sum_q ((x_q - x'_q) / lengthscales_q)^2)``."""

    _kind = _hip.KERNEL_RBF

    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        _Stationary.__init__(self, input_dim, variance, lengthscales, active_dims, ARD)

    def K(self, X, X2=None):
        """Gram matrix on the host (training-set side only)."""
        return self.variance * np.exp(-0.5 * self._square_dist(X, X2))


class Matern32(_Stationary):
    """``variance * (1 + sqrt(3) r) * exp(-sqrt(3) r)`` with gpflow 0.4.0's
    ``r = sqrt(square_dist + 1e-12)`` (``Stationary.euclid_dist``)."""

    _kind = _hip.KERNEL_MATERN32

    def K(self, X, X2=None):
        r = np.sqrt(3.0) * np.sqrt(self._square_dist(X, X2) + 1e-12)
        return self.variance * (1.0 + r) * np.exp(-r)


class Linear(Kern):
    """``k(x, x') = sum_q variance_q x_q x'_q`` (gpflow 0.4.0 ``kernels.Linear``; one variance
    for all dimensions unless ``ARD``)."""

    _kind = _hip.KERNEL_LINEAR

    def __init__(self, input_dim, variance=1.0, active_dims=None, ARD=False):
        Kern.__init__(self, input_dim, active_dims)
        self.ARD = bool(ARD)
        self.variance = np.broadcast_to(np.asarray(variance, dtype=config.np_dtype),
                                        (self.input_dim,)).copy()

    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        return (X * self.variance).dot(X2.T)

    def Kdiag(self, X):
        X, _ = self._slice(X, None)
        return np.sum(np.square(X) * self.variance, axis=1)

    def _fill(self, variance, inv_ls):
        variance[self.active_dims] = self.variance


class _Combination(Kern):
    def __init__(self, kern_list):
        self.kern_list = []
        for kern in kern_list:
            if not isinstance(kern, Kern):
                raise TypeError('can only combine Kern instances')
            # gpflow flattens nested combinations of the same kind
            self.kern_list.extend(kern.kern_list if type(kern) is type(self) else [kern])
        Kern.__init__(self, max(k.input_dim for k in self.kern_list))


class Add(_Combination):
    def K(self, X, X2=None):
        return sum(k.K(X, X2) for k in self.kern_list)

    def Kdiag(self, X):
        return sum(k.Kdiag(X) for k in self.kern_list)

    def _products(self):
        return [product for k in self.kern_list for product in k._products()]


class Prod(_Combination):
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

    def _products(self):
        products = [[]]
        for k in self.kern_list:                       # (a + b) * c = a c + b c
            products = [left + right for left in products for right in k._products()]
        return products


def _is_plain_rbf(kern, p):
    """The kernel of ``sl_gp_set_head`` (and of the MFMA paths that generate RBF values by
    recurrence): an RBF over all ``p`` inputs in their order."""
    return (type(kern) is RBF and kern.input_dim == p
            and np.array_equal(kern.active_dims, np.arange(p)))


class GPRCached(object):
    """GP regression model whose Cholesky data is cached for prediction
    (``functions.py:357-458``).  ``kern`` is a :class:`Kern`; ``mean_function`` a
    :class:`LinearSystem` (or ``None`` for zero mean); ``likelihood_variance`` is gpflow's
    ``likelihood.variance`` (default 1.0, the notebooks overwrite it).  ``scale`` is accepted for
    signature compatibility; it cancels analytically in ``build_predict`` and is not used."""

    def __init__(self, x, y, kern, mean_function=None, scale=1., name='GPRCached',
                 likelihood_variance=1.0):
        self.X = np.atleast_2d(np.asarray(x, dtype=config.np_dtype))
        self.Y = np.atleast_2d(np.asarray(y, dtype=config.np_dtype))
        if not isinstance(kern, Kern):
            raise TypeError('kern must be an RBF, Matern32 or Linear kernel or a sum / product of those')
        if mean_function is not None and not isinstance(mean_function, LinearSystem):
            raise TypeError('mean_function must be a LinearSystem or None')
        self.kern = kern
        self.mean_function = mean_function
        self._scale = float(scale)
        self.likelihood_variance = float(likelihood_variance)
        self.name = name
        self.update_cache()

    def update_cache(self):
        """Cholesky factor, its inverse and ``alpha = L^-1 (Y - m(X))`` (``functions.py:395-415``)."""
        n = len(self.X)
        gram = self.kern.K(self.X) + self.likelihood_variance * np.eye(n)
        self.cholesky = scipy.linalg.cholesky(gram, lower=True) if n else np.zeros((0, 0))
        resid = self.Y.copy()
        if self.mean_function is not None:
            resid = resid - self.X.dot(self.mean_function.matrix.T)
        if n == 0:                                      # no observations yet: the prior
            self.alpha, self.cholesky_inverse = np.zeros_like(resid), np.zeros((0, 0))
        else:
            self.alpha = scipy.linalg.solve_triangular(self.cholesky, resid, lower=True)
            self.cholesky_inverse = scipy.linalg.solve_triangular(self.cholesky, np.eye(n), lower=True)
            self.cholesky_inverse = np.tril(self.cholesky_inverse)
        self._version = next(_TOKENS)
        self._append_log = []          # rank-one extensions since the last full rebuild

    def append_data(self, x, y):
        """Add observations with a rank-one extension of the cached factors, O(n^2) per point
        instead of the reference's O(n^3) rebuild (``functions.py:525-546`` calls
        ``update_cache``).  For ``L' = [[L, 0], [l^T, s]]``: ``l = L^-1 k(X, x)``,
        ``s = sqrt(k(x, x) + sigma_n^2 - l.l)``, ``L'^-1 = [[L^-1, 0], [-(l^T L^-1)/s, 1/s]]`` and
        ``alpha' = [alpha; (r - l^T alpha)/s]`` with ``r = y - m(x)``."""
        x = np.atleast_2d(np.asarray(x, dtype=config.np_dtype))
        y = np.atleast_2d(np.asarray(y, dtype=config.np_dtype))
        for xi, yi in zip(x, y):
            xi, yi = xi[None, :], yi[None, :]
            n = len(self.X)
            k_vec = self.kern.K(self.X, xi)[:, 0]
            row = self.cholesky_inverse.dot(k_vec)                       # l = L^-1 k
            # K(x, x) as the rebuild of functions.py:401 forms it (kern.K, not kern.Kdiag: the
            # two differ by 1.5e-12 for a Matern32 leaf, whose distance carries euclid_dist's 1e-12)
            s2 = self.kern.K(xi)[0, 0] + self.likelihood_variance - row.dot(row)
            if not s2 > 0:
                raise np.linalg.LinAlgError('kernel matrix is not positive definite')
            s = np.sqrt(s2)
            resid = yi[0] - (xi.dot(self.mean_function.matrix.T)[0]
                             if self.mean_function is not None else 0.0)
            chol = np.zeros((n + 1, n + 1))
            chol[:n, :n], chol[n, :n], chol[n, n] = self.cholesky, row, s
            inv = np.zeros((n + 1, n + 1))
            inv[:n, :n] = self.cholesky_inverse
            inv[n, :n] = -row.dot(self.cholesky_inverse) / s
            inv[n, n] = 1.0 / s
            self.alpha = np.vstack((self.alpha, ((resid - row.dot(self.alpha)) / s)[None, :]))
            self.cholesky, self.cholesky_inverse = chol, inv
            self.X = np.vstack((self.X, xi))
            self.Y = np.vstack((self.Y, yi))
            # what the engine needs to follow without re-packing L^-1: the new row and alpha row
            before, self._version = self._version, next(_TOKENS)
            self._append_log.append((before, self._version, xi[0].copy(), inv[n, :n + 1].copy(),
                                     self.alpha[n].copy()))


class GaussianProcess(UncertainFunction):
    """``(mean, beta * sqrt(var))`` of a GP model (``functions.py:461-546``)."""

    _role = 'dynamics'

    def __init__(self, gaussian_process, beta=2., name='gaussian_process'):
        self.gaussian_process = gaussian_process
        self.beta = float(beta)
        self.input_dim = gaussian_process.X.shape[1]
        self.output_dim = gaussian_process.Y.shape[1]
        self.n_dim = self.input_dim
        self.name = name

    @property
    def X(self):
        return self.gaussian_process.X

    @property
    def Y(self):
        return self.gaussian_process.Y

    def add_data_point(self, x, y):
        """Append observations (``functions.py:525-546``); the cached factors are extended by a
        rank-one update instead of being rebuilt."""
        self.gaussian_process.append_data(x, y)


class FunctionStack(UncertainFunction):
    """One uncertain function per output column (``functions.py:254-307``)."""

    _role = 'dynamics'

    def __init__(self, functions, name='function_stack'):
        self.functions = list(functions)
        self.num_fun = len(self.functions)
        self.input_dim = self.functions[0].input_dim
        self.output_dim = sum(fun.output_dim for fun in self.functions)
        self.name = name

    def add_data_point(self, x, y):
        for fun, yi in zip(self.functions, np.asarray(y).squeeze()):
            fun.add_data_point(x, yi)


def _gp_heads(dynamics):
    """Flatten a GaussianProcess / FunctionStack into ``[(model, beta, col0)]``."""
    funs = dynamics.functions if isinstance(dynamics, FunctionStack) else [dynamics]
    heads, col = [], 0
    for fun in funs:
        if not isinstance(fun, GaussianProcess):
            raise TypeError('FunctionStack members must be GaussianProcess instances')
        heads.append((fun.gaussian_process, fun.beta, col))
        col += fun.output_dim
    betas = {b for _, b, _ in heads}
    if len(betas) != 1:
        raise ValueError('all GP heads must share the same beta')
    return heads, betas.pop()


# ----------------------------------------------------------------------------------------------
# piecewise-linear table on a grid
# ----------------------------------------------------------------------------------------------

class Triangulation(DeterministicFunction):
    """Delaunay interpolation of per-vertex values on a ``GridWorld`` (``functions.py:981-1510``).

    As in the reference only ONE unit cell is triangulated (SciPy/Qhull on the cell's corners,
    ``functions.py:1019-1022``) and reused for every cell.  ``parameters`` is the ``[nindex, k]``
    vertex table; the device copy is refreshed whenever it is assigned."""

    _role = 'value'

    def __init__(self, discretization, vertex_values=None, project=False, name='triangulation'):
        self.discretization = disc = discretization
        self.input_dim = disc.ndim
        self.project = bool(project)
        self.name = name
        self._parameters = None
        self._device_table = None
        self._table_thunk = None                   # (callable, columns): a device table not built yet
        self._table_version = next(_TOKENS)
        self._structure_token = next(_TOKENS)      # grid + unit-cell simplices never change
        if vertex_values is not None:
            self.parameters = vertex_values
        d = disc.ndim
        if d == 1:
            simplices = np.array([[0, 1]], dtype=np.int32)            # functions.py:935-958
        elif d in UNIT_CELL_SIMPLICES:
            # the reference asks Qhull (functions.py:1019-1023); which of the valid triangulations
            # of a box Qhull returns, and in which order, is frozen in unit_cells.py (the answer of
            # scipy 1.15.3 for every cell size): another SciPy cannot move product and oracle
            # together (tests/test_host_logic.py compares this table, the fixture and live SciPy)
            simplices = np.array(UNIT_CELL_SIMPLICES[d], dtype=np.int32)
        else:
            simplices = qhull_unit_cell(disc.unit_maxes)               # d > 4: parity unpinned
        self.unit_simplex_codes = simplices
        self.nsimplex_unit = len(simplices)
        self.nsimplex = self.nsimplex_unit * disc.nrectangles
        if self.nsimplex_unit > _hip.MAX_SIMPLICES:
            raise DimensionError('unit cell has %d simplices (engine limit %d)'
                                 % (self.nsimplex_unit, _hip.MAX_SIMPLICES))
        bits = (simplices[:, :, None] >> np.arange(d)) & 1           # [s, d+1, d]
        verts = bits.astype(config.np_dtype) * disc.unit_maxes
        self.hyperplanes = np.stack([np.linalg.inv(v[1:] - v[:1]) for v in verts])   # :1090-1101

    @property
    def nindex(self):
        return self.discretization.nindex

    @property
    def output_dim(self):
        if self._parameters is not None:
            return self._parameters.shape[1]
        if self._device_table is not None:
            return int(self._device_table.shape[1])
        if self._table_thunk is not None:
            return int(self._table_thunk[1])
        return None

    @property
    def parameters(self):
        """``[nindex, k]`` vertex table on the host; after a sweep that left the table on the GPU
        (value iteration, policy improvement) it is copied back on first access."""
        return self._host_parameters()

    @parameters.setter
    def parameters(self, values):
        self._parameters = np.ascontiguousarray(
            np.asarray(values, dtype=config.np_dtype).reshape(self.nindex, -1))
        self._device_table = None
        self._table_thunk = None
        self._table_version = next(_TOKENS)

    def _device(self, ctx):
        """Device copy of the vertex table (uploaded lazily)."""
        import torch
        self._resolve_table()
        if self._device_table is None or self._device_table.device != ctx.torch_device:
            self._device_table = torch.from_numpy(self._host_parameters()).to(ctx.torch_device)
        return self._device_table

    def _adopt_device_table(self, tensor):
        """Make a device tensor the truth (value iteration keeps V on the GPU)."""
        self._device_table = tensor.reshape(self.nindex, -1)
        self._table_thunk = None
        self._parameters = None
        self._table_version = next(_TOKENS)

    def _adopt_lazy_device_table(self, build, columns):
        """Like ``_adopt_device_table`` with a table that is only built (``build()`` -> device
        tensor) when somebody reads it: the greedy policy of a value-iteration sweep is replaced by
        the next sweep's before anything looked at it."""
        self._device_table = None
        self._table_thunk = (build, int(columns))
        self._parameters = None
        self._table_version = next(_TOKENS)

    def _resolve_table(self):
        if self._table_thunk is not None:
            build, _ = self._table_thunk
            self._table_thunk = None
            self._device_table = build().reshape(self.nindex, -1)

    def __getstate__(self):
        self._resolve_table()                      # a table that is not built yet is a closure: build it
        return self.__dict__

    def _host_parameters(self):
        self._resolve_table()
        if self._parameters is None and self._device_table is not None:
            self._parameters = self._device_table.cpu().numpy().reshape(self.nindex, -1)
        return self._parameters

    def _upload(self, ctx, slot):
        ctx.tri_set(slot, self.discretization._desc(), self.unit_simplex_codes, self.hyperplanes,
                    self.discretization.discrete_points, self.project,
                    self._device(ctx).shape[1], self._device(ctx))


# ----------------------------------------------------------------------------------------------
# analytic dynamics of the examples
# ----------------------------------------------------------------------------------------------

def _normalization(normalization):
    if normalization is None:
        return None, None
    norm = [np.array(v, dtype=config.np_dtype) for v in normalization]
    return norm, [v ** -1 for v in norm]


def _pendulum_linearize(mass, length, friction, dt, norm, gravity=9.81):
    """Discretised linearisation of the pendulum about the upright position
    (``examples/utilities.py:207-240``)."""
    inertia = mass * length ** 2
    A = np.array([[0, 1], [gravity / length, -friction / inertia]])
    B = np.array([[0], [1 / inertia]])
    Tx, Tu = np.diag(norm[0]), np.diag(norm[1])
    A = np.linalg.multi_dot((np.linalg.inv(Tx), A, Tx))
    B = np.linalg.multi_dot((np.linalg.inv(Tx), B, Tu))
    sysd = scipy.signal.StateSpace(A, B, np.eye(2), np.zeros((2, 1))).to_discrete(dt)
    return sysd.A, sysd.B


def _cartpole_linearize(m, M, L, b, dt, norm, gravity=9.81):
    """Zero-order-hold discretisation of the linearised cart-pole
    (``examples/utilities.py:352-385``)."""
    g = gravity
    A = np.array([[0, 0, 1, 0], [0, 0, 0, 1], [0, g * m / M, 0, -b / (M * L)],
                  [0, g * (m + M) / (L * M), 0, -b * (m + M) / (m * M * L ** 2)]])
    B = np.array([0, 0, 1 / M, 1 / (M * L)]).reshape(-1, 1)
    Tx, Tu = np.diag(norm[0]), np.diag(norm[1])
    A = np.linalg.multi_dot((np.linalg.inv(Tx), A, Tx))
    B = np.linalg.multi_dot((np.linalg.inv(Tx), B, Tu))
    Ad, Bd, _, _, _ = scipy.signal.cont2discrete((A, B, 0, 0), dt, method='zoh')
    return Ad, Bd


class InvertedPendulum(DeterministicFunction):
    """Pendulum with 10 explicit-Euler sub-steps (``examples/utilities.py:144-289``)."""

    _role = 'dynamics'

    def __init__(self, mass, length, friction=0, dt=1 / 80, normalization=None):
        self.mass, self.length, self.friction, self.dt = mass, length, friction, dt
        self.gravity = 9.81
        self.normalization, self.inv_norm = _normalization(normalization)
        self.input_dim, self.output_dim = 3, 2

    @property
    def inertia(self):
        return self.mass * self.length ** 2

    def linearize(self):
        """``(A, B)`` of the discretised linearisation about the upright rest position, in
        normalised coordinates when a normalisation is set (same result as
        ``examples/utilities.py:207-240``)."""
        norm = self.normalization if self.normalization is not None else [np.ones(2), np.ones(1)]
        return _pendulum_linearize(self.mass, self.length, self.friction, self.dt, norm,
                                   gravity=self.gravity)

    def _write_dynamics(self, desc):
        desc.kind = _hip.DYN_PENDULUM
        desc.coef[0] = self.dt / 10
        desc.coef[1] = self.gravity / self.length
        desc.coef[2] = self.inertia
        desc.coef[3] = self.friction / self.inertia
        desc.coef[4] = 1.0 if self.friction > 0 else 0.0
        _write_norm(desc, self.normalization, self.inv_norm, 2)


class CartPole(DeterministicFunction):
    """Cart-pole with 10 explicit-Euler sub-steps (``examples/utilities.py:292-437``)."""

    _role = 'dynamics'

    def __init__(self, pendulum_mass, cart_mass, length, rot_friction=0.0, dt=0.01,
                 normalization=None):
        self.pendulum_mass, self.cart_mass, self.length = pendulum_mass, cart_mass, length
        self.rot_friction, self.dt, self.gravity = rot_friction, dt, 9.81
        self.state_dim, self.action_dim = 4, 1
        self.input_dim, self.output_dim = 5, 4
        self.normalization, self.inv_norm = _normalization(normalization)

    def linearize(self):
        """``(A, B)`` of the zero-order-hold discretisation of the linearised cart-pole, in
        normalised coordinates when a normalisation is set (same result as
        ``examples/utilities.py:352-385``)."""
        norm = self.normalization if self.normalization is not None else [np.ones(4), np.ones(1)]
        return _cartpole_linearize(self.pendulum_mass, self.cart_mass, self.length,
                                   self.rot_friction, self.dt, norm, gravity=self.gravity)

    def _write_dynamics(self, desc):
        m, M, L, b, g = (self.pendulum_mass, self.cart_mass, self.length, self.rot_friction,
                         self.gravity)
        desc.kind = _hip.DYN_CARTPOLE
        # the products below are evaluated in the operator order of examples/utilities.py:430-433
        for k, v in enumerate([self.dt / 10, m, M, L, b, m * L, 0.5 * m * g * L, 0.5 * m * L,
                               b * (m + M), (m + M) * g]):
            desc.coef[k] = float(v)
        _write_norm(desc, self.normalization, self.inv_norm, 4)


def _write_norm(desc, normalization, inv_norm, d):
    desc.normalize = 0 if normalization is None else 1
    for k in range(d):
        desc.tx[k] = 1.0 if normalization is None else float(normalization[0][k])
        desc.tx_inv[k] = 1.0 if normalization is None else float(inv_norm[0][k])
    desc.tu[0] = 1.0 if normalization is None else float(normalization[1][0])


# ----------------------------------------------------------------------------------------------
# positive-definite network
# ----------------------------------------------------------------------------------------------

_ACTIVATIONS = {None: 0, 'linear': 0, 'tanh': 1, 'relu': 2}


class NeuralNetwork(DeterministicFunction):
    """A chain of dense layers as a POLICY (``functions.py:1663-1729``; the policy of
    ``examples/inverted_pendulum.ipynb:215`` and of the reinforcement-learning notebooks).

    ``layers`` are the units of every dense layer (``build_evaluation``, ``:1702-1725``: one
    ``tf.layers.dense`` per entry, the input width is the data's), ``nonlinearities`` one entry per
    layer - ``'relu'``, ``'tanh'``, ``'sigmoid'`` or ``None`` (the reference passes TensorFlow
    callables; the names of the ones it uses are accepted here) -, ``output_scale`` multiplies the
    output (``:1727-1728``); with ``use_bias`` every layer but the last has a bias (``:1712, 1722``).

    The reference's parameters are TensorFlow variables (Xavier initialisation, trained by SGD -
    outside this package's scope): here they are given, ``parameters = [W_0, b_0, W_1, b_1, ...,
    W_out]`` in the reference's variable order (``_parameter_iter``, ``:1731-1740``; ``W_l`` is
    ``[in, units]``; without biases just the ``W_l``), or drawn Xavier-uniform from ``seed`` once
    ``input_dim`` is known.  Assigning ``parameters`` again (or calling :meth:`touch` after editing the
    arrays in place) makes the engine upload them again."""

    _role = 'policy'
    _ACTIVATIONS = {None: 0, 'linear': 0, 'tanh': 1, 'relu': 2, 'sigmoid': 3}

    def __init__(self, layers, nonlinearities, output_scale=1., use_bias=True, name='neural_network',
                 input_dim=None, parameters=None, seed=0):
        self.layers = [int(v) for v in layers]
        self.nonlinearities = [getattr(a, '__name__', a) for a in nonlinearities]
        if len(self.nonlinearities) != len(self.layers):
            raise ValueError('one nonlinearity (or None) per layer')
        for a in self.nonlinearities:
            if a not in self._ACTIVATIONS:
                raise TypeError('nonlinearity %r: the engine has relu, tanh, sigmoid and None' % (a,))
        self.output_scale = float(output_scale)
        self.use_bias = bool(use_bias)
        self.name = name
        self.output_dim = self.layers[-1]
        self.input_dim = None if input_dim is None else int(input_dim)
        self._parameters = None
        self._version = next(_TOKENS)
        self._seed = seed
        if parameters is not None:
            self.parameters = parameters
        elif input_dim is not None:
            self._initialise()

    def _shapes(self):
        widths = [self.input_dim] + self.layers
        shapes = []
        for l in range(len(self.layers)):
            shapes.append((widths[l], widths[l + 1]))
            if self.use_bias and l < len(self.layers) - 1:
                shapes.append((widths[l + 1],))
        return shapes

    def _initialise(self):
        rng = np.random.default_rng(self._seed)
        params = []
        for shape in self._shapes():
            if len(shape) == 2:                       # tf.contrib.layers.xavier_initializer (uniform)
                params.append(rng.uniform(-1, 1, shape) * np.sqrt(6. / (shape[0] + shape[1])))
            else:
                params.append(np.zeros(shape))        # tf.layers.dense: zero bias
        self.parameters = params

    @property
    def parameters(self):
        return self._parameters

    @parameters.setter
    def parameters(self, values):
        values = [np.asarray(v, dtype=config.np_dtype) for v in values]
        if self.input_dim is None:
            self.input_dim = int(values[0].shape[0])
        if [v.shape for v in values] != self._shapes():
            raise ValueError('parameters must have shapes %s' % (self._shapes(),))
        self._parameters = values
        self._version = next(_TOKENS)

    def touch(self):
        """Say that the parameter arrays were edited in place."""
        self._version = next(_TOKENS)

    def _layers_for_upload(self, d):
        """``(dims, activation codes, kernels, biases)`` of the chain for a d-dimensional state."""
        if self.input_dim is None:
            self.input_dim = int(d)
            self._initialise()
        if self.input_dim != d:
            raise ValueError('the network expects %d inputs, the grid has %d dimensions'
                             % (self.input_dim, d))
        it = iter(self._parameters)
        kernels, biases = [], []
        for l in range(len(self.layers)):
            kernels.append(next(it))
            biases.append(next(it) if (self.use_bias and l < len(self.layers) - 1) else None)
        return ([self.input_dim] + self.layers, [self._ACTIVATIONS[a] for a in self.nonlinearities],
                kernels, biases)


class LyapunovNetwork(DeterministicFunction):
    """``sum(phi(x)^2)`` with layer kernels ``[W^T W + eps I ; W']``
    (``examples/utilities.py:48-104``).  ``activations`` are names ('tanh', 'relu', None);
    ``weights`` the flat variable list in the reference's creation order, or ``None`` for
    Xavier-uniform initialisation from ``seed``."""

    _role = 'value'

    def __init__(self, input_dim, layer_dims, activations, eps=1e-6, weights=None, seed=0,
                 name='lyapunov_network'):
        self.input_dim = int(input_dim)
        self.num_layers = len(layer_dims)
        self.activations = list(activations)
        self.eps = float(eps)
        self.name = name
        if layer_dims[0] < input_dim:
            raise ValueError('The first layer dimension must be at least the input dimension!')
        if not np.all(np.diff(layer_dims) >= 0):
            raise ValueError('Each layer must maintain or increase the dimension of its input!')
        self.output_dims = list(layer_dims)
        self.hidden_dims = [int(np.ceil(((self.input_dim if i == 0 else layer_dims[i - 1]) + 1) / 2))
                            for i in range(self.num_layers)]
        shapes = self.weight_shapes()
        if weights is None:
            rng = np.random.default_rng(seed)
            weights = [rng.uniform(-1, 1, s) * np.sqrt(6. / (s[0] + s[1])) for s in shapes]
        self.weights = [np.asarray(w, dtype=config.np_dtype) for w in weights]
        if [w.shape for w in self.weights] != shapes:
            raise ValueError('weights must have shapes %s' % (shapes,))

    def weight_shapes(self):
        shapes = []
        for i in range(self.num_layers):
            in_dim = self.input_dim if i == 0 else self.output_dims[i - 1]
            shapes.append((self.hidden_dims[i], in_dim))
            if self.output_dims[i] > in_dim:
                shapes.append((self.output_dims[i] - in_dim, in_dim))
        return shapes

    def kernels(self):
        """Per-layer ``[out_i, in_i]`` matrices (``examples/utilities.py:95-100``)."""
        out, it = [], iter(self.weights)
        for i in range(self.num_layers):
            in_dim = self.input_dim if i == 0 else self.output_dims[i - 1]
            W = next(it)
            kernel = W.T.dot(W) + self.eps * np.eye(in_dim)
            if self.output_dims[i] > in_dim:
                kernel = np.concatenate([kernel, next(it)], axis=0)
            out.append(kernel)
        return out

    def _upload(self, ctx):
        dims = [self.input_dim] + self.output_dims
        ctx.network_set(dims, [_ACTIVATIONS[a] for a in self.activations], self.kernels())
