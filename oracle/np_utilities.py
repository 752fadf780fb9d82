"""Oracle (test infrastructure): batch iteration and DLQR.

Follows ``safe_learning/utilities.py:224-249`` (batchify) and ``:327-357`` (dlqr).
"""

import numpy as np
import scipy.linalg


def batchify(arrays, batch_size):
    """Yield ``(start, [array[start:start+batch_size], ...])`` in order.

    Reference: ``safe_learning/utilities.py:224-249``.  The slices are views, the
    last batch may be short, iteration stops at the first empty batch.
    """
    if not isinstance(arrays, (list, tuple)):
        arrays = (arrays,)
    start = 0
    while True:
        batches = [array[start:start + batch_size] for array in arrays]
        if not batches[0].size:
            break
        yield start, batches
        start += batch_size


def dlqr(a, b, q, r):
    """Discrete-time LQR, ``u = -k x``.  Reference: ``safe_learning/utilities.py:327-357``."""
    a, b, q, r = map(np.atleast_2d, (a, b, q, r))
    p = scipy.linalg.solve_discrete_are(a, b, q, r)
    bp = b.T.dot(p)
    tmp1 = bp.dot(b)
    tmp1 = tmp1 + r
    tmp2 = bp.dot(a)
    k = np.linalg.solve(tmp1, tmp2)
    return k, p
