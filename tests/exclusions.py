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

Round 5: ambiguous points are no longer dropped.  At such a point the set of admissible answers
is finite - one value per unit-cell simplex that contains the wrapped point (``candidate_values``)
- and the reference returns one of them; the engine's answer has to be a MEMBER of that set
(``assert_member`` / ``check_own_vertices``), and where the reference run's own answer is on
record it is a member too.  What is logged for such points is the "ambiguous, membership-checked"
fraction; the ``own vertices`` exclusion limits are gone.

Every call is logged; ``conftest.py`` writes the log to ``gpurun_out/parity_exclusions.json`` at
the end of a GPU session.  ``SL_EXCLUSION_SOFT=1`` logs without asserting (to survey the rates).
"""

import os

import numpy as np

LOG = []
LIMITS = {"successor": 0.02}


def _simplex_candidates(otri, pts, eps):
    """Per unit-cell simplex s: does it contain the wrapped point (``functions.py:1116-1130``), and
    what does the table evaluate to through it (``:1160-1202, 1473-1499``) -> (inside[n, S],
    values[n, S, cols])."""
    disc = otri.discretization
    nsimp = otri.triangulation.nsimplex
    params = np.asarray(otri.parameters)
    unit = disc._center_states(pts, clip=True) % disc.unit_maxes
    rect = disc.state_to_rectangle(pts)
    eval_pts = np.clip(pts, disc.limits[:, 0], disc.limits[:, 1]) if otri.project else pts
    inside = np.zeros((len(pts), nsimp), dtype=bool)
    values = np.full((len(pts), nsimp, params.shape[1]), np.nan)
    for s in range(nsimp):
        verts = disc.index_to_state(otri.unit_simplices[s]) - disc.offset
        w1 = (unit - verts[0]).dot(otri.hyperplanes[s])
        w0 = 1 - w1.sum(axis=1)
        inside[:, s] = (w1 >= -eps).all(axis=1) & (w0 >= -eps)
        simplices = otri.simplices(s + rect * nsimp)
        origins = disc.index_to_state(simplices[:, 0])
        w = (eval_pts - origins).dot(otri.hyperplanes[s])
        w = np.hstack((1 - w.sum(axis=1, keepdims=True), w))
        values[:, s, :] = np.einsum("nk,nkc->nc", w, params[simplices])
    return inside, values


def ambiguous_points(otri, pts, eps=1e-11, rtol=1e-10):
    """Points at which the reference's interpolated value depends on scipy's search history.

    The reference locates a point with unit-cell coordinates ``(x - offset) % unit_maxes``
    (functions.py:1116-1124).  When these lie on a face shared by several unit-cell simplices,
    scipy's ``find_simplex`` returns whichever simplex its walk reaches first (it starts from the
    previous query's result).  That is harmless when the candidates agree on the value, but the
    ``%`` wrap-around can put the unit coordinates in a different corner than the true position;
    the candidates then extrapolate differently and the reference's value is history dependent.
    Such points (several candidates, disagreeing values in any output column) have no single
    right answer: see ``candidate_values``."""
    disc = otri.discretization
    if disc.ndim == 1:
        return np.zeros(len(pts), dtype=bool)
    inside, values = _simplex_candidates(otri, np.asarray(pts, dtype=np.float64), eps)
    masked = np.where(inside[:, :, None], values, np.nan)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            lo, hi = np.nanmin(masked, axis=1), np.nanmax(masked, axis=1)
    spread = (hi - lo) > rtol * np.maximum(1.0, np.maximum(np.abs(lo), np.abs(hi)))
    return np.nan_to_num(spread, nan=0.0).astype(bool).any(axis=1)


def candidate_values(otri, pts, eps=1e-11):
    """The admissible answers of the reference's table look-up at ``pts``: ``values[n, S, cols]``
    with row s = the table evaluated through unit-cell simplex s, NaN where s does not contain the
    wrapped point.  SciPy returns ONE of the containing simplices (which one depends on where its
    walk started), so the reference's value is one of the finite rows - and so must the engine's be."""
    inside, values = _simplex_candidates(otri, np.asarray(pts, dtype=np.float64), eps)
    return np.where(inside[:, :, None], values, np.nan)


def is_member(got, candidates, rtol=1e-9, atol=1e-12):
    """``got[n, cols]`` equals one of ``candidates[n, S, cols]`` (all columns of the same row)."""
    got = np.asarray(got, dtype=np.float64).reshape(len(candidates), 1, -1)
    with np.errstate(invalid="ignore"):
        close = np.abs(candidates - got) <= atol + rtol * np.abs(candidates)
    return close.all(axis=2).any(axis=1)


def report_membership(test, ambiguous, member, kind="own vertices"):
    """Log the ambiguous fraction of a comparison and assert that every ambiguous point's answer is
    one of the admissible ones (nothing is excluded)."""
    ambiguous, member = np.asarray(ambiguous, dtype=bool), np.asarray(member, dtype=bool)
    entry = {"test": test, "kind": kind + ": ambiguous, membership-checked", "points": int(ambiguous.size),
             "ambiguous": float(ambiguous.mean()) if ambiguous.size else 0.0,
             "members": int(member.sum()), "checked": int(member.size), "excluded": 0.0}
    LOG.append(entry)
    print("parity [%s] %s: %.4f of %d points ambiguous, %d of %d answers admissible" % (
        kind, test, entry["ambiguous"], ambiguous.size, member.sum(), member.size))
    assert member.all(), "%s: %d of %d ambiguous points got a value that no containing simplex yields" % (
        test, int((~member).sum()), member.size)


def _mean(next_states):
    return next_states[0] if isinstance(next_states, tuple) else next_states


def admissible_future_values(orl, opolicy, states, eps=1e-11, **kwargs):
    """``[n, S]`` future values ``r(x, u_s) + gamma V(f(x, u_s))`` (``reinforcement_learning.py:
    65-114``) for every admissible policy value ``u_s`` at ``states`` (NaN where simplex s does not
    contain the wrapped state), and ``[n]``: some candidate's successor is itself an ambiguous
    point of the VALUE table (no single right answer for that candidate; category "successor").
    ``kwargs`` go to the oracle's ``future_values`` (lyapunov, ...)."""
    cands = candidate_values(opolicy, states, eps)                 # [n, S, m]
    out = np.full(cands.shape[:2], np.nan)
    successor = np.zeros(len(states), dtype=bool)
    for s in range(cands.shape[1]):
        rows = np.isfinite(cands[:, s, :]).all(axis=1)
        if rows.any():
            out[rows, s] = np.asarray(orl.future_values(states[rows], actions=cands[rows, s, :],
                                                        **kwargs)).reshape(-1)
            successor[rows] |= ambiguous_points(orl.value_function,
                                                _mean(orl.dynamics(states[rows], cands[rows, s, :])))
    return out, successor


def check_own_vertices(test, orl, opolicy, states, got, also=None, rtol=1e-9, atol=1e-12, **kwargs):
    """The engine's future values ``got`` at states where the POLICY table is ambiguous: each has to
    be the future value under one of the admissible policy values; ``also`` (the reference run's own
    numbers, when on record) must be admissible too.  Call it while the oracle's tables are those
    the engine's sweep read.  Returns the ambiguity mask (the caller compares the other points for
    equality)."""
    states = np.asarray(states, dtype=np.float64)
    amb = ambiguous_points(opolicy, states, eps=1e-11)
    if amb.any():
        cands, successor = admissible_future_values(orl, opolicy, states[amb], **kwargs)
        cands = cands[:, :, None]
        member = is_member(np.asarray(got).reshape(len(states), -1)[amb], cands, rtol, atol) | successor
        if also is not None:
            ref_member = is_member(np.asarray(also).reshape(len(states), -1)[amb], cands, rtol, atol)
            assert (ref_member | successor).all(), \
                "%s: the reference run's own value is not among the candidates" % test
        report(test + " (successors of the candidates)", ~successor, "successor")
    else:
        member = np.zeros(0, dtype=bool)
    report_membership(test, amb, member)
    return amb


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
