"""Host-side helpers kept from the reference's utilities (``safe_learning/utilities.py``)."""

import numpy as np
import scipy.linalg

__all__ = ['dlqr', 'batchify']


def dlqr(a, b, q, r):
    """Discrete-time LQR gain and cost-to-go, ``u = -k x`` (``utilities.py:327-357``)."""
    a, b, q, r = (np.atleast_2d(v) for v in (a, b, q, r))
    p = scipy.linalg.solve_discrete_are(a, b, q, r)
    k = np.linalg.solve(b.T.dot(p).dot(b) + r, b.T.dot(p).dot(a))
    return k, p


def batchify(arrays, batch_size):
    """Yield ``(start, [views])`` over consecutive batches (``utilities.py:224-249``)."""
    if not isinstance(arrays, (list, tuple)):
        arrays = (arrays,)
    start = 0
    while arrays[0][start:start + batch_size].size:
        yield start, [a[start:start + batch_size] for a in arrays]
        start += batch_size
