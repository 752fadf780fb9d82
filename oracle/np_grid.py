"""Oracle (test infrastructure): the regular state grid.

Follows ``safe_learning/functions.py:579-817`` (``GridWorld``) formula by formula.
"""

import numpy as np

_EPS = np.finfo(np.float64).eps


class DimensionError(Exception):
    """Reference: ``safe_learning/functions.py:575-576``."""


class GridWorld(object):
    """Regular grid.  Reference: ``safe_learning/functions.py:591-620``."""

    def __init__(self, limits, num_points):
        self.limits = np.atleast_2d(limits).astype(np.float64)
        num_points = np.broadcast_to(num_points, len(self.limits))
        self.num_points = num_points.astype(np.int64, copy=False)
        if np.any(self.num_points < 2):
            raise DimensionError('There must be at least 2 points in each dimension.')
        self.offset = self.limits[:, 0]
        self.unit_maxes = ((self.limits[:, 1] - self.offset)
                           / (self.num_points - 1)).astype(np.float64)
        self.offset_limits = np.stack((np.zeros_like(self.limits[:, 0]),
                                       self.limits[:, 1] - self.offset), axis=1)
        self.discrete_points = [np.linspace(low, up, n, dtype=np.float64)
                                for (low, up), n in zip(self.limits, self.num_points)]
        self.nrectangles = int(np.prod(self.num_points - 1))
        self.nindex = int(np.prod(self.num_points))
        self.ndim = len(self.limits)
        self._all_points = None

    @property
    def all_points(self):
        """All grid points, C order (last dim fastest).  Reference: ``functions.py:622-638``."""
        if self._all_points is None:
            mesh = np.meshgrid(*self.discrete_points, indexing='ij')
            points = np.column_stack([col.ravel() for col in mesh])
            self._all_points = points.astype(np.float64)
        return self._all_points

    def __len__(self):
        return self.nindex

    def _check_dimensions(self, states):
        """Reference: ``functions.py:679-689``."""
        if not states.shape[1] == self.ndim:
            raise DimensionError('the input argument has the wrong dimensions.')

    def _center_states(self, states, clip=True):
        """Reference: ``functions.py:691-712``."""
        states = np.atleast_2d(states).astype(np.float64)
        states = states - self.offset[None, :]
        if clip:
            np.clip(states, self.offset_limits[:, 0] + 2 * _EPS,
                    self.offset_limits[:, 1] - 2 * _EPS, out=states)
        return states

    def index_to_state(self, indices):
        """``ijk * unit_maxes + offset`` (multiply, then add).  Reference: ``functions.py:714-731``."""
        indices = np.atleast_1d(indices)
        ijk_index = np.vstack(np.unravel_index(indices, self.num_points)).T
        ijk_index = ijk_index.astype(np.float64)
        return ijk_index * self.unit_maxes + self.offset

    def state_to_index(self, states):
        """Nearest grid index (clip, round-half-even).  Reference: ``functions.py:733-752``."""
        states = np.atleast_2d(states)
        self._check_dimensions(states)
        states = np.clip(states, self.limits[:, 0], self.limits[:, 1])
        states = (states - self.offset) * (1. / self.unit_maxes)
        ijk_index = np.rint(states).astype(np.int32)
        return np.ravel_multi_index(ijk_index.T, self.num_points)

    def state_to_rectangle(self, states):
        """Reference: ``functions.py:754-776``."""
        ind = []
        for i, (discrete, num_points) in enumerate(zip(self.discrete_points, self.num_points)):
            idx = np.digitize(states[:, i], discrete)
            idx -= 1
            np.clip(idx, 0, num_points - 2, out=idx)
            ind.append(idx)
        return np.ravel_multi_index(ind, self.num_points - 1)

    def rectangle_to_state(self, rectangles):
        """Reference: ``functions.py:778-798``."""
        rectangles = np.atleast_1d(rectangles)
        ijk_index = np.vstack(np.unravel_index(rectangles, self.num_points - 1))
        ijk_index = ijk_index.astype(np.float64)
        return (ijk_index.T * self.unit_maxes) + self.offset

    def rectangle_corner_index(self, rectangles):
        """Reference: ``functions.py:800-817``."""
        ijk_index = np.vstack(np.unravel_index(rectangles, self.num_points - 1))
        return np.ravel_multi_index(np.atleast_2d(ijk_index), self.num_points)
