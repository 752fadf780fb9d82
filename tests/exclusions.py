"""Book-keeping of what the parity tests of the table look-ups EXCLUDE, and a bound on it.

The reference locates a point of a ``Triangulation`` with unit-cell coordinates
``(x - offset) % unit_maxes`` and SciPy's ``find_simplex`` (``functions.py:1103-1130``), whose walk
starts at the previous query's answer.  Where several unit-cell simplices contain the wrapped
point AND extrapolate to different values, the reference's own result depends on the query history
(``tests/test_gpu_rl.py::ambiguous_points``); parity is undefined there and such points are left
out of the comparison.  Two categories, measured on the oracle side (tools/exclusion_rates.py):

* ``successor`` - the value table read at successor states ``f(x, u)``, inside the grid or
  projected onto its boundary faces (``project=True``): measured 0 of 1e5 points in every test
  shape, 8 ... 87 % of them ON a boundary face.  Projected successors are therefore COMPARED, not
  excluded; the bound is 2 %.  On the GPU suite (profiles/r04_parity_exclusions.json, 171 logged
  comparisons): at most 0.23 %; the C5 sweep at 64^4 compares 2503 sampled vertices x 9 actions
  with NONE excluded, 61 % of them having a successor on a boundary face.
* ``own vertices`` - a piecewise-constant (greedy) policy table evaluated at its own vertices
  (``reinforcement_learning.py:98``: ``policy(states)`` at the grid points): every query sits on
  grid lines in all dimensions, the ``%`` wraps it into a corner of the unit cell shared by several
  simplices, and a noisy table makes them disagree: 19 ... 36 % of the vertices of the small random
  test tables (41 % on the 13 x 13 table of a reference-run scenario), 0 % for smooth tables, 2.0 %
  for the greedy policy of C5 at 64^4.  Bound 45 % on the small random tables, 5 % at full size.

Every call is logged; ``conftest.py`` writes the log to ``gpurun_out/parity_exclusions.json`` at
the end of a GPU session.  ``SL_EXCLUSION_SOFT=1`` logs without asserting (to survey the rates).
"""

import os

import numpy as np

LOG = []
LIMITS = {"successor": 0.02, "own vertices": 0.45, "own vertices, full size": 0.05}


def on_boundary_face(tri, points):
    """Points that ``project=True`` clips onto a boundary face of the table's grid."""
    limits = np.asarray(tri.discretization.limits, dtype=np.float64)
    return ((points <= limits[:, 0]) | (points >= limits[:, 1])).any(axis=1)


def report(test, ok, kind, limit=None, faces=None, soft=False):
    """Log and bound the excluded fraction ``1 - mean(ok)`` of one comparison.  ``faces``: mask of
    the compared points that lie on a projected boundary face (logged; they must not be excluded
    wholesale)."""
    ok = np.asarray(ok, dtype=bool)
    limit = LIMITS[kind] if limit is None else limit
    excluded = 1.0 - float(ok.mean()) if ok.size else 0.0
    entry = {"test": test, "kind": kind, "points": int(ok.size), "excluded": excluded, "limit": limit}
    if faces is not None:
        faces = np.asarray(faces, dtype=bool)
        entry["on_boundary_faces"] = float(faces.mean())
        entry["on_boundary_faces_compared"] = float((faces & ok.reshape(faces.shape)).sum() / max(faces.sum(), 1))
    LOG.append(entry)
    print("parity exclusions [%s] %s: %.4f of %d points (limit %.2f)%s" % (
        kind, test, excluded, ok.size, limit,
        "" if faces is None else "; %.3f on boundary faces, %.3f of those compared"
        % (entry["on_boundary_faces"], entry["on_boundary_faces_compared"])))
    if os.environ.get("SL_EXCLUSION_SOFT") != "1" and not soft:
        assert excluded <= limit, "%s: %.3f of the points excluded from parity (%s, limit %.2f)" % (
            test, excluded, kind, limit)
    return excluded


def within(test, ok, kind, limit=None):
    """``report`` for worker processes that collect failures instead of raising -> bool."""
    limit = LIMITS[kind] if limit is None else limit
    return report(test, ok, kind, limit, soft=True) <= limit or os.environ.get("SL_EXCLUSION_SOFT") == "1"
