"""Oracle (test infrastructure): function objects of the Lyapunov sweep, NumPy float64.

Every class is a callable ``f(*points) -> ndarray[B, out]`` (uncertain functions return
``(mean, error_bound)``), mirroring how the reference composes TensorFlow callables.
Reference files followed (relative to the upstream checkout):

* ``safe_learning/functions.py:254-291``  FunctionStack
* ``safe_learning/functions.py:310-354``  Saturation
* ``safe_learning/functions.py:357-458``  GPRCached (cache + predict)
* ``safe_learning/functions.py:461-546``  GaussianProcess (beta * sqrt(var), add_data_point)
* ``safe_learning/functions.py:981-1326`` _Triangulation
* ``safe_learning/functions.py:1513-1583`` QuadraticFunction, LinearSystem
* ``examples/utilities.py:48-104``   LyapunovNetwork
* ``examples/utilities.py:144-289``  InvertedPendulum
* ``examples/utilities.py:292-437``  CartPole
* gpflow==0.4.0 ``kernels.py`` (``Stationary.square_dist`` / ``euclid_dist``, ``RBF``, ``Matern32``,
  ``Linear``, ``Add``, ``Prod``: ``K`` and ``Kdiag``) and
  ``gpr.py`` (``GPR.build_predict``): not in the checkout, published algorithm restated.
"""

from itertools import product as cartesian

import numpy as np
import scipy.linalg
from scipy import spatial, signal


# --------------------------------------------------------------------------------------
# Canonical small linear algebra (see oracle/__init__.py, "Canonical arithmetic").
# --------------------------------------------------------------------------------------

def _hstack_inputs(points):
    """``concatenate_inputs``: ``safe_learning/utilities.py:123-159``."""
    arrays = [np.atleast_2d(np.asarray(p, dtype=np.float64)) for p in points]
    if len(arrays) == 1:
        return arrays[0]
    return np.hstack(arrays)


def ordered_matmul(x, matrix):
    """``x[B,k] @ matrix[k,o]`` accumulated left to right, one rounding per mul and per add."""
    x = np.asarray(x, dtype=np.float64)
    matrix = np.asarray(matrix, dtype=np.float64)
    nrows, ninner = x.shape
    out = np.empty((nrows, matrix.shape[1]), dtype=np.float64)
    for j in range(matrix.shape[1]):
        acc = x[:, 0] * matrix[0, j]
        for i in range(1, ninner):
            acc = acc + x[:, i] * matrix[i, j]
        out[:, j] = acc
    return out


def ordered_rowsum(x):
    """Sum over axis 1, left to right, keepdims."""
    acc = x[:, 0].copy()
    for i in range(1, x.shape[1]):
        acc = acc + x[:, i]
    return acc[:, None]


# --------------------------------------------------------------------------------------
# Deterministic building blocks
# --------------------------------------------------------------------------------------

class QuadraticFunction(object):
    """``x P x^T`` per row; P need not be symmetric.  Reference: ``functions.py:1513-1543``."""

    def __init__(self, matrix):
        self.matrix = np.atleast_2d(matrix).astype(np.float64)
        self.ndim = self.matrix.shape[0]

    def __call__(self, *points):
        points = _hstack_inputs(points)
        linear_form = ordered_matmul(points, self.matrix)        # functions.py:1537
        return ordered_rowsum(linear_form * points)              # functions.py:1538-1539

    def gradient(self, points):
        """``x (P + P^T)``.  Reference: ``functions.py:1541-1543``."""
        return ordered_matmul(np.atleast_2d(points), self.matrix + self.matrix.T)


class LinearSystem(object):
    """``[x, u] M^T`` with ``M = hstack(matrices)``.  Reference: ``functions.py:1546-1583``."""

    def __init__(self, matrices):
        if isinstance(matrices, np.ndarray):
            matrices = (matrices,)
        self.matrix = np.hstack([np.atleast_2d(m).astype(np.float64) for m in matrices])
        self.output_dim, self.input_dim = self.matrix.shape

    def __call__(self, *points):
        return ordered_matmul(_hstack_inputs(points), self.matrix.T)


class Saturation(object):
    """Clamp the output of ``fun``.  Reference: ``functions.py:310-354``."""

    def __init__(self, fun, lower, upper):
        self.fun, self.lower, self.upper = fun, lower, upper

    def __call__(self, *points):
        return np.minimum(np.maximum(self.fun(*points), self.lower), self.upper)


class AbsFunction(object):
    """``|fun(x)|`` - the notebooks' ``L_v = lambda x: tf.abs(grad(x))``
    (``examples/adaptive_safety_verification.ipynb`` cell 17)."""

    def __init__(self, fun):
        self.fun = fun

    def __call__(self, *points):
        return np.abs(self.fun(*points))


class Norm1Function(object):
    """``||fun(x)||_1`` per row, keepdims (same notebook cell, non-scaling branch)."""

    def __init__(self, fun):
        self.fun = fun

    def __call__(self, *points):
        return ordered_rowsum(np.abs(self.fun(*points)))


class NegatedFunction(object):
    """``-f`` as ``MultipliedFunction(f, -1)``.  Reference: ``functions.py:120-122, 196-199``."""

    def __init__(self, fun):
        self.fun = fun

    def __call__(self, *points):
        return self.fun(*points) * -1.0


class ConstantPolicy(object):
    """The same action row for every state (``reinforcement_learning.py:233-235, 268``)."""

    def __init__(self, action):
        self.action = np.atleast_1d(np.asarray(action, dtype=np.float64))

    def __call__(self, states):
        return np.broadcast_to(self.action, (len(np.atleast_2d(states)), len(self.action))).copy()


# --------------------------------------------------------------------------------------
# Gaussian process (gpflow 0.4.0 arithmetic restated)
# --------------------------------------------------------------------------------------

class RBF(object):
    """Squared-exponential kernel as in gpflow==0.4.0 ``kernels.RBF``.

    ``K(X, X2) = variance * exp(-square_dist(X, X2) / 2)`` with
    ``square_dist = -2 (X/l)(X2/l)^T + |X/l|^2 + |X2/l|^2`` (``Stationary.square_dist``) and
    ``Kdiag(X) = variance``.  Pinned by ``safe_learning/tests/test_functions.py:237-261``.
    """

    def __init__(self, input_dim, variance=1.0, lengthscales=None, ARD=False):
        self.input_dim = int(input_dim)
        self.variance = float(variance)
        if lengthscales is None:
            lengthscales = np.ones(self.input_dim) if ARD else 1.0
        self.lengthscales = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64),
                                            (self.input_dim,)).copy()
        self.ARD = ARD

    def square_dist(self, X, X2=None):
        X = X / self.lengthscales
        Xs = np.sum(np.square(X), 1)
        if X2 is None:
            return -2 * X.dot(X.T) + Xs[:, None] + Xs[None, :]
        X2 = X2 / self.lengthscales
        X2s = np.sum(np.square(X2), 1)
        return -2 * X.dot(X2.T) + Xs[:, None] + X2s[None, :]

    def K(self, X, X2=None):
        return self.variance * np.exp(-self.square_dist(X, X2) / 2)

    def Kdiag(self, X):
        return np.full(len(X), self.variance, dtype=np.float64)


class _Sliced(object):
    """``Kern._slice`` of gpflow==0.4.0: a kernel reads the columns ``active_dims`` of its inputs
    (default: the first ``input_dim``), and ``+`` / ``*`` build ``Add`` / ``Prod``."""

    def _init_dims(self, input_dim, active_dims):
        self.input_dim = int(input_dim)
        self.active_dims = (np.arange(self.input_dim) if active_dims is None
                            else np.asarray(list(active_dims), dtype=np.int64))

    def _slice(self, X, X2):
        X = np.asarray(X, dtype=np.float64)[:, self.active_dims]
        return X, (None if X2 is None else np.asarray(X2, dtype=np.float64)[:, self.active_dims])

    def __add__(self, other):
        return Add([self, other])

    def __mul__(self, other):
        return Prod([self, other])


class Matern32(_Sliced):
    """gpflow==0.4.0 ``kernels.Matern32``: ``variance (1 + sqrt(3) r) exp(-sqrt(3) r)`` with
    ``r = Stationary.euclid_dist = sqrt(square_dist + 1e-12)``; ``Kdiag = variance``.  Used by the
    reference's notebooks (``examples/inverted_pendulum.ipynb:152-158``), pinned by none of its
    tests: parity unpinned beyond the published formula."""

    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        self._init_dims(input_dim, active_dims)
        self._rbf = RBF(input_dim, variance, lengthscales, ARD)      # square_dist, lengthscales
        self.variance, self.lengthscales = self._rbf.variance, self._rbf.lengthscales

    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        r = np.sqrt(self._rbf.square_dist(X, X2) + 1e-12)
        return self.variance * (1. + np.sqrt(3.) * r) * np.exp(-np.sqrt(3.) * r)

    def Kdiag(self, X):
        return np.full(len(X), self.variance, dtype=np.float64)


class SlicedRBF(_Sliced):
    """``kernels.RBF`` with ``active_dims`` (the plain :class:`RBF` above reads every column)."""

    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        self._init_dims(input_dim, active_dims)
        self._rbf = RBF(input_dim, variance, lengthscales, ARD)
        self.variance, self.lengthscales = self._rbf.variance, self._rbf.lengthscales

    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        return self._rbf.K(X, X2)

    def Kdiag(self, X):
        return self._rbf.Kdiag(X)


class Linear(_Sliced):
    """gpflow==0.4.0 ``kernels.Linear``: ``K = (X * variance) X2^T``, ``Kdiag = sum(X^2 * variance)``
    (one variance per active dimension with ``ARD``, else one for all)."""

    def __init__(self, input_dim, variance=1.0, active_dims=None, ARD=False):
        self._init_dims(input_dim, active_dims)
        self.variance = np.broadcast_to(np.asarray(variance, dtype=np.float64), (self.input_dim,)).copy()

    def K(self, X, X2=None):
        X, X2 = self._slice(X, X2)
        return (X * self.variance).dot((X if X2 is None else X2).T)

    def Kdiag(self, X):
        X, _ = self._slice(X, None)
        return np.sum(np.square(X) * self.variance, 1)


class Add(_Sliced):
    """gpflow==0.4.0 ``kernels.Add``: the sum of the members' ``K`` / ``Kdiag``."""

    def __init__(self, kern_list):
        self.kern_list = list(kern_list)

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
    """gpflow==0.4.0 ``kernels.Prod``: the elementwise product."""

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


class GPRCached(object):
    """GP regression with cached Cholesky factor.  Reference: ``functions.py:357-458``.

    ``likelihood_variance`` defaults to gpflow's Gaussian likelihood default of 1.0 (the
    notebooks overwrite ``gp.likelihood.variance`` after construction).
    """

    def __init__(self, x, y, kern, mean_function=None, scale=1., likelihood_variance=1.0):
        self.X = np.atleast_2d(np.asarray(x, dtype=np.float64))
        self.Y = np.atleast_2d(np.asarray(y, dtype=np.float64))
        self.kern = kern
        self.mean_function = mean_function
        self._scale = float(scale)
        self.likelihood_variance = float(likelihood_variance)
        self.update_cache()

    def _mean(self, X):
        if self.mean_function is None:                       # gpflow.mean_functions.Zero
            return np.zeros((len(X), self.Y.shape[1]), dtype=np.float64)
        return self.mean_function(X)

    def update_cache(self):
        """Reference: ``functions.py:395-415``."""
        n = len(self.X)
        kernel = self.kern.K(self.X) + np.eye(n) * self.likelihood_variance   # :401
        kernel = kernel * (self._scale ** 2)                                  # :402
        target = self._scale * (self.Y - self._mean(self.X))                  # :405
        self.cholesky = scipy.linalg.cholesky(kernel, lower=True)             # :408
        self.alpha = scipy.linalg.solve_triangular(self.cholesky, target, lower=True)  # :409

    def build_predict(self, Xnew):
        """Posterior mean and marginal variance.  Reference: ``functions.py:417-458``."""
        Xnew = np.atleast_2d(np.asarray(Xnew, dtype=np.float64))
        Kx = (self._scale ** 2) * self.kern.K(self.X, Xnew)                   # :438
        mx = self._scale * self._mean(Xnew)                                   # :439
        a = scipy.linalg.solve_triangular(self.cholesky, Kx, lower=True)      # :441
        fmean = a.T.dot(self.alpha) + mx                                      # :442
        Knew = (self._scale ** 2) * self.kern.Kdiag(Xnew)                     # :450
        fvar = Knew - np.sum(np.square(a), 0)                                 # :451
        fvar = np.tile(fvar.reshape(-1, 1), [1, self.Y.shape[1]])             # :452
        fmean = fmean / self._scale                                           # :455
        fvar = fvar / (self._scale ** 2)                                      # :456
        return fmean, fvar


class GaussianProcess(object):
    """``(mean, beta * sqrt(var))``.  Reference: ``functions.py:461-546``."""

    def __init__(self, gaussian_process, beta=2.):
        self.gaussian_process = gaussian_process
        self.beta = float(beta)
        self.input_dim = gaussian_process.X.shape[1]
        self.output_dim = gaussian_process.Y.shape[1]

    @property
    def X(self):
        return self.gaussian_process.X

    @property
    def Y(self):
        return self.gaussian_process.Y

    def __call__(self, *points):
        mean, var = self.gaussian_process.build_predict(_hstack_inputs(points))   # :512
        return mean, self.beta * np.sqrt(var)                                     # :514

    def add_data_point(self, x, y):
        """Reference: ``functions.py:525-546`` (full cache rebuild)."""
        gp = self.gaussian_process
        gp.X = np.vstack((gp.X, np.atleast_2d(x)))
        gp.Y = np.vstack((gp.Y, np.atleast_2d(y)))
        gp.update_cache()


class FunctionStack(object):
    """Independent uncertain functions, one per output column.  Reference: ``functions.py:254-307``."""

    def __init__(self, functions):
        self.functions = list(functions)
        self.input_dim = self.functions[0].input_dim
        self.output_dim = sum(fun.output_dim for fun in self.functions)

    def __call__(self, *points):
        means, errors = [], []
        for fun in self.functions:
            mean, error = fun(*points)
            means.append(mean)
            errors.append(error)
        return np.concatenate(means, axis=1), np.concatenate(errors, axis=1)

    def add_data_point(self, x, y):
        for fun, yi in zip(self.functions, np.asarray(y).squeeze()):
            fun.add_data_point(x, yi)


# --------------------------------------------------------------------------------------
# Piecewise-linear interpolation on the grid
# --------------------------------------------------------------------------------------

class _Delaunay1D(object):
    """Reference: ``functions.py:935-978``."""

    def __init__(self, points):
        self.points = points
        self.nsimplex = len(points) - 1
        self._min, self._max = np.min(points), np.max(points)
        self.simplices = np.array([[0, 1]])

    def find_simplex(self, points):
        points = points.squeeze()
        out_of_bounds = (points > self._max) | (points < self._min)
        return np.where(out_of_bounds, -1, 0)


class Triangulation(object):
    """Delaunay interpolation with one triangulated unit cell.  Reference: ``functions.py:981-1326``."""

    def __init__(self, discretization, vertex_values=None, project=False):
        self.discretization = disc = discretization
        self.input_dim = disc.ndim
        self._parameters = None
        self.parameters = vertex_values
        if len(disc.limits) == 1:                                             # :1015-1017
            corners = np.array([[0], disc.unit_maxes])
            self.triangulation = _Delaunay1D(corners)
        else:                                                                 # :1019-1022
            corners = np.array(list(cartesian(*np.diag(disc.unit_maxes))), dtype=np.float64)
            self.triangulation = spatial.Delaunay(corners)
        self.unit_simplices = self._triangulation_simplex_indices()
        self.nsimplex = self.triangulation.nsimplex * disc.nrectangles
        self.hyperplanes = None
        self._update_hyperplanes()
        self.project = project

    @property
    def nindex(self):
        return self.discretization.nindex

    @property
    def parameters(self):
        return self._parameters

    @parameters.setter
    def parameters(self, values):
        if values is None:
            self._parameters = None
        else:
            self._parameters = np.asarray(values, dtype=np.float64).reshape(self.nindex, -1)

    @property
    def output_dim(self):
        return None if self._parameters is None else self._parameters.shape[1]

    def _triangulation_simplex_indices(self):
        """Reference: ``functions.py:1064-1088``."""
        disc = self.discretization
        simplices = self.triangulation.simplices
        new_simplices = np.empty_like(simplices)
        index_mapping = disc.state_to_index(self.triangulation.points + disc.offset)
        for i, new_index in enumerate(index_mapping):
            new_simplices[simplices == i] = new_index
        return new_simplices

    def _update_hyperplanes(self):
        """Reference: ``functions.py:1090-1101``."""
        nsimp = self.triangulation.nsimplex
        self.hyperplanes = np.empty((nsimp, self.input_dim, self.input_dim), dtype=np.float64)
        for i, simplex in enumerate(self.unit_simplices):
            pts = self.discretization.index_to_state(simplex)
            self.hyperplanes[i] = np.linalg.inv(pts[1:] - pts[:1])

    def find_simplex(self, points):
        """Reference: ``functions.py:1103-1130``."""
        disc = self.discretization
        rectangles = disc.state_to_rectangle(points)
        centred = disc._center_states(points, clip=True)
        unit_coordinates = centred % disc.unit_maxes
        simplex_ids = np.atleast_1d(self.triangulation.find_simplex(unit_coordinates))
        simplex_ids = simplex_ids + rectangles * self.triangulation.nsimplex
        return simplex_ids

    def simplices(self, indices):
        """Reference: ``functions.py:1132-1158``."""
        unit_indices = np.remainder(indices, self.triangulation.nsimplex)
        simplices = self.unit_simplices[unit_indices].copy()
        rectangles = np.floor_divide(indices, self.triangulation.nsimplex)
        corner_index = self.discretization.rectangle_corner_index(rectangles)
        if simplices.ndim > 1:
            corner_index = corner_index[:, None]
        simplices += corner_index
        return simplices

    def _get_weights(self, points):
        """Barycentric weights.  Reference: ``functions.py:1160-1202`` / ``:1473-1491``."""
        disc = self.discretization
        simplex_ids = self.find_simplex(points)
        simplices = self.simplices(simplex_ids)
        origins = disc.index_to_state(simplices[:, 0])
        hyperplanes = self.hyperplanes[simplex_ids % self.triangulation.nsimplex]
        if self.project:
            points = np.clip(points, disc.limits[:, 0], disc.limits[:, 1])
        weights = np.empty((len(points), self.input_dim + 1), dtype=np.float64)
        offset = points - origins
        np.sum(offset[:, :, None] * hyperplanes, axis=1, out=weights[:, 1:])
        weights[:, 0] = 1 - np.sum(weights[:, 1:], axis=1)
        return weights, simplices

    def __call__(self, *points):
        points = np.atleast_2d(_hstack_inputs(points))
        weights, simplices = self._get_weights(points)
        return np.sum(weights[:, :, None] * self.parameters[simplices], axis=1)   # :1223-1226

    def gradient(self, points):
        """Reference: ``functions.py:1261-1326``."""
        points = np.atleast_2d(points)
        simplex_ids = self.find_simplex(points)
        simplices = self.simplices(simplex_ids)
        simplex_ids = simplex_ids % self.triangulation.nsimplex
        weights = np.empty((len(simplex_ids), self.input_dim, self.input_dim + 1))
        weights[:, :, 1:] = self.hyperplanes[simplex_ids]
        weights[:, :, 0] = -np.sum(weights[:, :, 1:], axis=2)
        res = np.einsum('ijk,ikl->ilj', weights, self.parameters[simplices, :])
        if res.shape[1] == 1:
            res = res.squeeze(axis=1)
        return res


class TriangulationGradient(object):
    """``lambda x: tf.abs(tri.gradient(x))`` (``examples/inverted_pendulum.ipynb`` cell 14)."""

    def __init__(self, tri, absolute=True):
        self.tri, self.absolute = tri, absolute

    def __call__(self, points):
        grad = self.tri.gradient(points)
        return np.abs(grad) if self.absolute else grad


# --------------------------------------------------------------------------------------
# Analytic dynamics of the examples (10 explicit Euler sub-steps)
# --------------------------------------------------------------------------------------

class InvertedPendulum(object):
    """Reference: ``examples/utilities.py:144-289``."""

    def __init__(self, mass, length, friction=0, dt=1 / 80, normalization=None):
        self.mass, self.length, self.friction, self.dt = mass, length, friction, dt
        self.gravity = 9.81
        self.normalization = normalization
        if normalization is not None:
            self.normalization = [np.array(norm, dtype=np.float64) for norm in normalization]
            self.inv_norm = [norm ** -1 for norm in self.normalization]

    @property
    def inertia(self):
        return self.mass * self.length ** 2

    def linearize(self):
        """Reference: ``examples/utilities.py:207-240``."""
        A = np.array([[0, 1], [self.gravity / self.length, -self.friction / self.inertia]],
                     dtype=np.float64)
        B = np.array([[0], [1 / self.inertia]], dtype=np.float64)
        if self.normalization is not None:
            Tx, Tu = map(np.diag, self.normalization)
            Tx_inv, Tu_inv = map(np.diag, self.inv_norm)
            A = np.linalg.multi_dot((Tx_inv, A, Tx))
            B = np.linalg.multi_dot((Tx_inv, B, Tu))
        sysd = signal.StateSpace(A, B, np.eye(2), np.zeros((2, 1))).to_discrete(self.dt)
        return sysd.A, sysd.B

    def __call__(self, *state_action):
        sa = _hstack_inputs(state_action)
        state, action = sa[:, :2].copy(), sa[:, 2:3].copy()
        if self.normalization is not None:                                    # :194-205
            state = state * self.normalization[0]
            action = action * self.normalization[1]
        n_inner = 10
        dt = self.dt / n_inner
        for _ in range(n_inner):                                              # :249-253
            state = state + dt * self.ode(state, action)
        if self.normalization is not None:
            state = state * self.inv_norm[0]
        return state

    def ode(self, state, action):
        """Reference: ``examples/utilities.py:257-289``."""
        angle, angular_velocity = state[:, [0]], state[:, [1]]
        x_ddot = self.gravity / self.length * np.sin(angle) + action / self.inertia
        if self.friction > 0:
            x_ddot = x_ddot - self.friction / self.inertia * angular_velocity
        return np.concatenate((angular_velocity, x_ddot), axis=1)


class CartPole(object):
    """Reference: ``examples/utilities.py:292-437``."""

    def __init__(self, pendulum_mass, cart_mass, length, rot_friction=0.0, dt=0.01,
                 normalization=None):
        self.pendulum_mass, self.cart_mass, self.length = pendulum_mass, cart_mass, length
        self.rot_friction, self.dt, self.gravity = rot_friction, dt, 9.81
        self.state_dim, self.action_dim = 4, 1
        self.normalization = normalization
        if normalization is not None:
            self.normalization = [np.array(norm, dtype=np.float64) for norm in normalization]
            self.inv_norm = [norm ** -1 for norm in self.normalization]

    def linearize(self):
        """Reference: ``examples/utilities.py:352-385``."""
        m, M, L, b, g = (self.pendulum_mass, self.cart_mass, self.length, self.rot_friction,
                         self.gravity)
        A = np.array([[0, 0, 1, 0],
                      [0, 0, 0, 1],
                      [0, g * m / M, 0, -b / (M * L)],
                      [0, g * (m + M) / (L * M), 0, -b * (m + M) / (m * M * L ** 2)]],
                     dtype=np.float64)
        B = np.array([0, 0, 1 / M, 1 / (M * L)]).reshape((-1, self.action_dim))
        if self.normalization is not None:
            Tx, Tu = map(np.diag, self.normalization)
            Tx_inv, Tu_inv = map(np.diag, self.inv_norm)
            A = np.linalg.multi_dot((Tx_inv, A, Tx))
            B = np.linalg.multi_dot((Tx_inv, B, Tu))
        Ad, Bd, _, _, _ = signal.cont2discrete((A, B, 0, 0), self.dt, method='zoh')
        return Ad, Bd

    def __call__(self, *state_action):
        sa = _hstack_inputs(state_action)
        state, action = sa[:, :4].copy(), sa[:, 4:5].copy()
        if self.normalization is not None:
            state = state * self.normalization[0]
            action = action * self.normalization[1]
        inner_euler_steps = 10
        dt = self.dt / inner_euler_steps
        for _ in range(inner_euler_steps):                                    # :394-398
            state = state + dt * self.ode(state, action)
        if self.normalization is not None:
            state = state * self.inv_norm[0]
        return state

    def ode(self, state, action):
        """Reference: ``examples/utilities.py:402-437`` (operator order kept)."""
        m, M, L, b, g = (self.pendulum_mass, self.cart_mass, self.length, self.rot_friction,
                         self.gravity)
        theta, v, omega = state[:, [1]], state[:, [2]], state[:, [3]]
        sin_t, cos_t, sin_2t = np.sin(theta), np.cos(theta), np.sin(2 * theta)
        omega_sq = np.square(omega)
        det = L * (M + m * np.square(sin_t))
        v_dot = (action - m * L * omega_sq * sin_t - b * omega * cos_t
                 + 0.5 * m * g * L * sin_2t) * L / det
        omega_dot = (action * cos_t - 0.5 * m * L * omega_sq * sin_2t
                     - b * (m + M) * omega / (m * L) + (m + M) * g * sin_t) / det
        return np.concatenate((v, omega, v_dot, omega_dot), axis=1)


# --------------------------------------------------------------------------------------
# Positive-definite network of lyapunov_function_learning.ipynb
# --------------------------------------------------------------------------------------

class NeuralNetwork(object):
    """Chain of dense layers, ``net <- act_l(net W_l + b_l)``, output times ``output_scale``.

    Reference: ``functions.py:1663-1729`` (``build_evaluation``: one ``tf.layers.dense`` per entry of
    ``layers``, bias on every layer but the last when ``use_bias``, ``tf.multiply(net,
    output_scale)``).  The reference's parameters are TensorFlow variables; here they are given in its
    variable order ``[W_0, b_0, ..., W_out]`` (``_parameter_iter``, ``:1731-1740``).  No test of the
    reference holds a number for this class (``tests/test_functions.py:764-777`` checks that the
    Lipschitz constant is positive): the dense-layer arithmetic is restated, **parity unpinned**."""

    _ACT = {None: lambda x: x, 'linear': lambda x: x, 'tanh': np.tanh,
            'relu': lambda x: np.maximum(x, 0.0), 'sigmoid': lambda x: 1.0 / (1.0 + np.exp(-x))}

    def __init__(self, layers, nonlinearities, output_scale=1., use_bias=True, parameters=None):
        self.layers = list(layers)
        self.nonlinearities = list(nonlinearities)
        self.output_scale = output_scale
        self.use_bias = use_bias
        self.parameters = [np.asarray(p, dtype=np.float64) for p in parameters]
        self.input_dim = self.parameters[0].shape[0]
        self.output_dim = self.layers[-1]

    def __call__(self, points):
        net = np.atleast_2d(np.asarray(points, dtype=np.float64))
        it = iter(self.parameters)
        for l, activation in enumerate(self.nonlinearities):
            net = net.dot(next(it))                                         # :1708-1713 / :1717-1722
            if self.use_bias and l < len(self.layers) - 1:
                net = net + next(it)
            net = self._ACT[activation](net)
        return net * self.output_scale                                      # :1727


class LyapunovNetwork(object):
    """``sum(phi(x)^2)`` with layer kernels ``[W^T W + eps I ; W']``.

    Reference: ``examples/utilities.py:48-104``.  ``weights`` is a flat list in the order the
    reference creates its variables: per layer ``weights_posdef_i`` ``[hidden_i, in_i]`` then,
    when the layer widens, ``weights_i`` ``[out_i - in_i, in_i]``.
    """

    def __init__(self, input_dim, layer_dims, activations, eps=1e-6, weights=None):
        self.input_dim = int(input_dim)
        self.num_layers = len(layer_dims)
        self.activations = list(activations)
        self.eps = float(eps)
        if layer_dims[0] < input_dim:
            raise ValueError('The first layer dimension must be at least the input dimension!')
        if not np.all(np.diff(layer_dims) >= 0):
            raise ValueError('Each layer must maintain or increase the dimension of its input!')
        self.output_dims = list(layer_dims)
        self.hidden_dims = np.zeros(self.num_layers, dtype=int)
        for i in range(self.num_layers):
            layer_input_dim = self.input_dim if i == 0 else self.output_dims[i - 1]
            self.hidden_dims[i] = np.ceil((layer_input_dim + 1) / 2).astype(int)
        self.weights = [np.asarray(w, dtype=np.float64) for w in weights]

    def weight_shapes(self):
        shapes = []
        for i in range(self.num_layers):
            in_dim = self.input_dim if i == 0 else self.output_dims[i - 1]
            shapes.append((int(self.hidden_dims[i]), in_dim))
            if self.output_dims[i] - in_dim > 0:
                shapes.append((self.output_dims[i] - in_dim, in_dim))
        return shapes

    def kernels(self):
        """Per-layer ``kernel`` matrices ``[out_i, in_i]`` (``examples/utilities.py:95-100``)."""
        kernels, it = [], iter(self.weights)
        for i in range(self.num_layers):
            in_dim = self.input_dim if i == 0 else self.output_dims[i - 1]
            W = next(it)
            kernel = W.T.dot(W) + self.eps * np.eye(in_dim)
            if self.output_dims[i] - in_dim > 0:
                kernel = np.concatenate([kernel, next(it)], axis=0)
            kernels.append(kernel)
        return kernels

    @staticmethod
    def _act(name, x):
        if name == 'tanh':
            return np.tanh(x)
        if name == 'relu':
            return np.maximum(x, 0.)
        if name in (None, 'linear'):
            return x
        raise ValueError(name)

    @staticmethod
    def _dact(name, pre, post):
        if name == 'tanh':
            return 1. - post * post
        if name == 'relu':
            return (pre > 0).astype(np.float64)
        return np.ones_like(pre)

    def __call__(self, *points):
        net = _hstack_inputs(points)
        for kernel, act in zip(self.kernels(), self.activations):
            net = self._act(act, net.dot(kernel.T))                           # :101-102
        return np.sum(np.square(net), axis=1, keepdims=True)                  # :103

    def gradient(self, points):
        """d value / d input (what ``tf.gradients`` returns in
        ``examples/lyapunov_function_learning.ipynb`` cell with ``L_v``)."""
        net = np.atleast_2d(np.asarray(points, dtype=np.float64))
        kernels, cache = self.kernels(), []
        for kernel, act in zip(kernels, self.activations):
            pre = net.dot(kernel.T)
            post = self._act(act, pre)
            cache.append((pre, post))
            net = post
        grad = 2. * net
        for kernel, act, (pre, post) in zip(reversed(kernels), reversed(self.activations),
                                            reversed(cache)):
            grad = (grad * self._dact(act, pre, post)).dot(kernel)
        return grad
