"""CPU check of the kernels' per-cell arithmetic (safe_learning_amd/csrc/sl_model.h).

The header is compiled with g++ into a test-only shim (tests/hostsim) and compared with the
oracle: bit-for-bit for the linear / quadratic pipeline (canonical arithmetic), to rounding for
the Euler dynamics (libm sin/cos differ from NumPy's by an ulp).  The GPU tests repeat the same
comparisons through the real kernels.
"""

import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases
import oracle
from conftest import ROOT
from safe_learning_amd import functions as F
from safe_learning_amd._model import ModelBuilder
from safe_learning_amd.benchmarks import build_specs


class _RecordingCtx(object):
    """Captures the model description instead of uploading it (no GPU needed)."""
    torch_device = None

    def model_set(self, desc):
        self.desc = desc


@pytest.fixture(scope="module")
def hostsim():
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    lib = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    deps = [src, os.path.join(ROOT, "safe_learning_amd", "csrc", "sl_model.h"),
            os.path.join(ROOT, "include", "sl_hip.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "safe_learning_amd", "csrc"),
                               "-o", lib, src])
    h = C.CDLL(lib)
    h.hs_vbits.restype = C.c_uint64
    h.hs_vbits.argtypes = [C.c_double]
    h.hs_vbits_to_double.restype = C.c_double
    h.hs_vbits_to_double.argtypes = [C.c_uint64]
    return h


def _describe(case):
    grid = F.GridWorld(case['limits'], case['num_points'])
    policy, dynamics, value, lv = build_specs(case)
    ctx = _RecordingCtx()
    ModelBuilder(ctx, grid).upload(policy, dynamics, value, lv, case['lf'], case['tau'])
    return grid, ctx.desc


def _run(hostsim, case):
    grid, desc = _describe(case)
    n, d = grid.nindex, grid.ndim
    values = np.zeros(n)
    negative = np.zeros(n, dtype=np.uint8)
    dbg = np.zeros((n, 2 + 2 * d))
    rc = hostsim.hs_det_cells(C.byref(desc), C.c_int64(0), C.c_int64(n),
                              values.ctypes.data_as(C.c_void_p), negative.ctypes.data_as(C.c_void_p),
                              dbg.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return values, negative.astype(bool), dbg


CASES_EXACT = [
    ("1d", dict()),
    ("1d", dict(tau_scale=0.0)),
    ("1d", dict(num_points=3)),
    ("pendulum", dict(num_points=33, dynamics="linear", tau_scale=0.0)),
    ("pendulum", dict(num_points=[17, 40], dynamics="linear", tau_scale=0.02)),
    ("cartpole", dict(num_points=8, dynamics="linear", tau_scale=0.0)),
    ("cartpole", dict(num_points=[5, 6, 7, 9], dynamics="linear", tau_scale=0.01)),
]


@pytest.mark.parametrize("name,kw", CASES_EXACT)
def test_linear_pipeline_bit_exact(hostsim, name, kw):
    case = cases.make_case(name, **kw)
    values, negative, dbg = _run(hostsim, case)
    lyap = cases.oracle_lyapunov(case)
    idx = np.arange(lyap.discretization.nindex)
    rec = cases.oracle_cell_records(lyap, idx)
    assert_array_equal(values, lyap.values)                       # V(x) bit for bit
    assert_array_equal(dbg, rec)                                  # decrease, threshold, f(x)
    assert_array_equal(negative, lyap.negative(lyap.discretization.index_to_state(idx)))


@pytest.mark.parametrize("name,kw", [
    ("pendulum", dict(num_points=40, dynamics="analytic", tau_scale=0.0)),
    ("pendulum", dict(num_points=40, dynamics="analytic", tau_scale=0.02)),
    ("cartpole", dict(num_points=9, dynamics="analytic", tau_scale=0.0)),
])
def test_euler_dynamics(hostsim, name, kw):
    case = cases.make_case(name, **kw)
    values, negative, dbg = _run(hostsim, case)
    lyap = cases.oracle_lyapunov(case)
    idx = np.arange(lyap.discretization.nindex)
    rec = cases.oracle_cell_records(lyap, idx)
    assert_array_equal(values, lyap.values)
    assert_allclose(dbg, rec, rtol=1e-12, atol=1e-15)
    ref_neg = lyap.negative(lyap.discretization.index_to_state(idx))
    margin = np.abs(rec[:, 0] - rec[:, 1])
    differs = negative != ref_neg
    # a flipped bit is only acceptable if the cell sits within rounding of the threshold
    assert not np.any(differs & (margin > 1e-12 * np.maximum(1.0, np.abs(rec[:, 1]))))
    assert differs.sum() <= 2


def test_vbits_order(hostsim):
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.normal(size=200), [0.0, -0.0, np.inf, -np.inf, 1e-310, -1e-310, np.nan]])
    bits = np.array([hostsim.hs_vbits(float(x)) for x in v], dtype=np.uint64)
    order_bits = np.argsort(bits, kind="stable")
    order_val = np.argsort(v, kind="stable")          # NaN last, -0 == +0
    assert_array_equal(order_bits, order_val)
    back = np.array([hostsim.hs_vbits_to_double(int(b)) for b in bits])
    assert_array_equal(np.isnan(back), np.isnan(v))
    assert_array_equal(back[~np.isnan(v)], np.where(v[~np.isnan(v)] == 0, 0.0, v[~np.isnan(v)]))


def _tri_eval(hostsim, tri, points, col=0, want_grad=False):
    grid = tri.discretization
    desc = grid._desc()
    pts = np.ascontiguousarray(points, dtype=np.float64)
    out = np.zeros(len(pts))
    grad = np.zeros((len(pts), grid.ndim)) if want_grad else None
    simp = np.ascontiguousarray(tri.unit_simplex_codes, dtype=np.int32)
    hyper = np.ascontiguousarray(tri.hyperplanes)
    dp = np.ascontiguousarray(np.concatenate(grid.discrete_points))
    table = np.ascontiguousarray(tri.parameters)
    rc = hostsim.hs_tri_eval(C.byref(desc), len(simp), simp.ctypes.data_as(C.c_void_p),
                             hyper.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p),
                             int(tri.project), table.shape[1], table.ctypes.data_as(C.c_void_p),
                             C.c_int64(len(pts)), pts.ctypes.data_as(C.c_void_p), col,
                             out.ctypes.data_as(C.c_void_p),
                             grad.ctypes.data_as(C.c_void_p) if want_grad else None)
    assert rc == 0
    return (out, grad) if want_grad else out


@pytest.mark.parametrize("limits,num,project", [
    ([[0, 1]], [3], False),
    ([[-1, 1], [-1, 2]], [3, 7], False),
    ([[-1, 1], [-1, 2]], [3, 7], True),
    ([[0, 1]] * 3, [4, 3, 5], True),
    ([[-1, 1]] * 4, [4, 5, 3, 4], True),
    ([[-1, 1]] * 4, [4, 5, 3, 4], False),
])
@pytest.mark.parametrize("col", [0, -1], ids=["reference-order", "bellman-fast"])
def test_triangulation_matches_oracle(hostsim, limits, num, project, col):
    rng = np.random.default_rng(1)
    grid = F.GridWorld(limits, num)
    values = rng.normal(size=(grid.nindex, 1))
    tri = F.Triangulation(grid, values, project=project)
    ogrid = oracle.GridWorld(limits, num)
    otri = oracle.Triangulation(ogrid, values, project=project)
    span = np.diff(grid.limits, axis=1).T
    pts = [grid.all_points,                                          # vertices
           rng.uniform(0, 1, (400, grid.ndim)) * span + grid.offset]
    if project:
        # outside points: only with projection - without it the reference extrapolates with
        # whichever unit-cell simplex Qhull's walk returns for a point on a shared vertex
        pts.append(rng.uniform(-0.3, 1.3, (100, grid.ndim)) * span + grid.offset)
    pts = np.vstack(pts)
    got = _tri_eval(hostsim, tri, pts, col=col)
    assert_allclose(got[:, None], otri(pts), rtol=1e-10, atol=1e-12)


def test_point_location_shortcut(hostsim, monkeypatch):
    """The coordinate-order regions of sl_tri_regions: every 4-D region lists a handful of the 22
    unit-cell simplices, and the pruned walk returns the full walk's value bit for bit, also with
    the linspace points loaded instead of recomputed."""
    rng = np.random.default_rng(3)
    grid = F.GridWorld([[-1, 1]] * 4, [5, 4, 6, 3])
    tri = F.Triangulation(grid, rng.normal(size=(grid.nindex, 1)), project=True)
    ncand = _tri_eval(hostsim, tri, np.zeros((64, 4)), col=-2).astype(int)
    assert (ncand > 0).sum() == 24 and ncand.max() <= 8 and ncand[ncand > 0].min() >= 4
    span = np.diff(grid.limits, axis=1).T
    pts = np.vstack([rng.uniform(-0.2, 1.2, (4000, 4)) * span + grid.offset, grid.all_points,
                     0.5 * (grid.all_points[:-1] + grid.all_points[1:])])
    fast = _tri_eval(hostsim, tri, pts, col=-1)
    # second level (signs of z_i + z_j - 1): at most four candidates, and the candidate phase
    # settles every point that is not on a face of the unit cell in two or more coordinates
    probe = _tri_eval(hostsim, tri, pts, col=-3).astype(int)
    resolved, fine, tried = (probe & 1) == 1, (probe & 2) == 2, probe >> 2
    assert fine.all() and tried.max() <= 4 and tried.mean() < 3.5
    generic = pts[:4000]
    inside = (np.abs(generic - grid.offset) < span).all(axis=1) & (generic > grid.offset).all(axis=1)
    assert resolved[:4000][inside].all()
    monkeypatch.setenv("SL_HOSTSIM_NO_REGIONS", "1")
    assert_array_equal(fast, _tri_eval(hostsim, tri, pts, col=-1))
    monkeypatch.setenv("SL_HOSTSIM_LOAD_POINTS", "1")
    assert_array_equal(fast, _tri_eval(hostsim, tri, pts, col=-1))
    monkeypatch.delenv("SL_HOSTSIM_NO_REGIONS")
    grid3 = F.GridWorld([[0, 1]] * 3, [4, 3, 5])
    tri3 = F.Triangulation(grid3, rng.normal(size=(grid3.nindex, 1)))
    n3 = _tri_eval(hostsim, tri3, np.zeros((64, 3)), col=-2).astype(int)
    assert (n3 > 0).sum() == 6 and n3.max() == 1


def test_region_tables_are_remembered_per_dimension(hostsim):
    """sl_tri_regions keeps one result per dimension: a 2-D table in between must neither disturb
    nor be disturbed by the 4-D tables (same values before and after, sampled level included)."""
    rng = np.random.default_rng(11)
    g4 = F.GridWorld([[-1, 1]] * 4, [4, 5, 3, 4])
    t4 = F.Triangulation(g4, rng.normal(size=(g4.nindex, 1)), project=True)
    g2 = F.GridWorld([[-1, 1], [0, 3]], [9, 6])
    t2 = F.Triangulation(g2, rng.normal(size=(g2.nindex, 1)), project=True)
    p4 = rng.uniform(-1.1, 1.1, (3000, 4))
    p2 = rng.uniform(-1.1, 3.1, (3000, 2))
    first4 = _tri_eval(hostsim, t4, p4, col=-1)
    first2 = _tri_eval(hostsim, t2, p2, col=-1)
    assert_array_equal(_tri_eval(hostsim, t4, p4, col=-1), first4)
    assert_array_equal(_tri_eval(hostsim, t2, p2, col=-1), first2)
    assert_array_equal(_tri_eval(hostsim, t4, np.zeros((64, 4)), col=-2),
                       _tri_eval(hostsim, t4, np.zeros((64, 4)), col=-2))
    ot4 = oracle.Triangulation(oracle.GridWorld([[-1, 1]] * 4, [4, 5, 3, 4]), t4.parameters, project=True)
    assert_allclose(first4[:, None], ot4(p4), rtol=1e-10, atol=1e-12)


def test_fmod_exact(hostsim):
    """sl_fmod_exact (one fma with the integer quotient) returns numpy's `%` bit for bit."""
    hostsim.hs_fmod_exact.restype = C.c_double
    hostsim.hs_fmod_exact.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.uniform(1e-3, 3, 2000), [0.1, 0.125, 1 / 3, 2 / 63, 1e-9]])
    a = np.concatenate([rng.uniform(0, 50, 2000), [0.3, 0.375, 1.0, 62 * (2 / 63), 1.0]])
    # exact multiples and their neighbours, where the rounded quotient is off by one
    k = rng.integers(0, 200, len(b))
    hostsim.hs_fmod_exact_inv.restype = C.c_double
    hostsim.hs_fmod_exact_inv.argtypes = [C.c_double, C.c_double]
    for aa in (a, k * b, np.nextafter(k * b, np.inf), np.nextafter(k * b, 0)):
        aa = np.abs(aa)
        got = np.array([hostsim.hs_fmod_exact(float(x), float(y)) for x, y in zip(aa, b)])
        assert_array_equal(got, np.fmod(aa, b))
        # the lookups' variant: quotient guessed with the reciprocal, corrected by selects
        got = np.array([hostsim.hs_fmod_exact_inv(float(x), float(y)) for x, y in zip(aa, b)])
        assert_array_equal(got, np.fmod(aa, b))


def test_rectangle_index_matches_digitize(hostsim):
    """The branch-free rectangle search (one corrected guess) against the oracle's np.digitize
    on grid points, their floating-point neighbours, points outside the grid and NaN."""
    rng = np.random.default_rng(9)
    for limits, num in (([[-1, 1]] * 2, [64, 7]), ([[0.3, 2.9], [-5, -1], [0, 1e-3]], [33, 128, 5]),
                        ([[-1, 1]] * 4, [64, 64, 64, 64])):
        grid = F.GridWorld(limits, num)
        ogrid = oracle.GridWorld(limits, num)
        tri = F.Triangulation(grid, np.zeros((grid.nindex, 1)))
        d = grid.ndim
        cols = []
        for k in range(d):
            p = grid.discrete_points[k]
            cols.append(np.concatenate([p, np.nextafter(p, np.inf), np.nextafter(p, -np.inf),
                                        [p[0] - 1.0, p[-1] + 1.0, 1e300, -1e300, np.nan],
                                        rng.uniform(p[0] - 0.1, p[-1] + 0.1, 300)]))
        m = min(len(c) for c in cols)
        pts = np.stack([rng.permutation(c)[:m] for c in cols], axis=1)
        got = _tri_eval(hostsim, tri, pts, col=-4)
        want = ogrid.state_to_rectangle(pts)
        assert_array_equal(got, want)



def test_triangulation_golden_cases(hostsim, golden):
    """The reference's own literal cases through the kernel code path."""
    g = golden["triangulation_1d"]
    tri = F.Triangulation(F.GridWorld(g["limits"], g["num_points"]), g["vertex_values"])
    pts = np.array(g["test_points"], dtype=float)
    val, grad = _tri_eval(hostsim, tri, pts, want_grad=True)
    assert_allclose(val[:, None], g["expected_values"], atol=1e-15)
    interior = [1, 3, 4]                                             # away from the kinks
    assert_allclose(grad[interior], np.array(g["expected_gradient"])[interior])

    g = golden["triangulation_values"]
    p = g["projection"]
    grid = F.GridWorld(g["limits"], g["num_points"])
    tri = F.Triangulation(grid, p["parameters"])
    assert_allclose(_tri_eval(hostsim, tri, np.array(p["point"])), np.ravel(p["unprojected"]))
    tri.project = True
    assert_allclose(_tri_eval(hostsim, tri, np.array(p["point"])), np.ravel(p["projected"]))

    g = golden["triangulation_3d"]
    grid = F.GridWorld(g["limits"], g["num_points"])
    tri = F.Triangulation(grid, np.sum(grid.index_to_state(np.arange(8)), axis=1) / 3)
    assert tri.nsimplex == g["nsimplex"]
    assert_allclose(_tri_eval(hostsim, tri, np.array(g["test_points"], dtype=float)),
                    g["expected"], atol=g["atol"])

    g = golden["triangulation_gradient"]
    grid = F.GridWorld(g["limits"], g["num_points"])
    values = np.zeros(grid.nindex)
    values[grid.state_to_index(np.array(g["node_states"], dtype=float))] = g["node_values"]
    tri = F.Triangulation(grid, values)
    _, grad = _tri_eval(hostsim, tri, np.array(g["test_points"]), want_grad=True)
    assert_allclose(grad, g["expected_gradient"])


def test_exp_nonpos(hostsim):
    """sl_exp_nonpos (the 20-instruction exp of the Bellman kernels) within 2 ulp of numpy on the
    range the RBF kernel uses, exact at 0, monotone into the underflow."""
    hostsim.hs_exp_nonpos.restype = C.c_double
    hostsim.hs_exp_nonpos.argtypes = [C.c_double]
    rng = np.random.default_rng(7)
    x = -np.concatenate([rng.uniform(0, 50, 4000), rng.uniform(0, 1e-3, 500), rng.uniform(50, 745, 500),
                         [0.0, np.log(2) / 2, np.log(2), 700.0]])
    got = np.array([hostsim.hs_exp_nonpos(float(v)) for v in x])
    ref = np.exp(x)
    normal = ref > 1e-300
    assert np.max(np.abs(got[normal] - ref[normal]) / ref[normal]) < 4.5e-16
    assert_allclose(got[~normal], ref[~normal], rtol=1e-12, atol=5e-324)
    assert hostsim.hs_exp_nonpos(0.0) == 1.0
    assert hostsim.hs_exp_nonpos(-800.0) == 0.0
    assert hostsim.hs_exp_nonpos(-np.inf) == 0.0
    assert np.isnan(hostsim.hs_exp_nonpos(np.nan))


def test_sincos(hostsim):
    """sl_sincos (shared reduction, fdlibm kernels) within 2 ulp of numpy over the angles the Euler
    integrators see, quadrant handling included."""
    hostsim.hs_sincos.restype = None
    hostsim.hs_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.uniform(-10, 10, 6000), rng.uniform(-1e4, 1e4, 1000),
                        np.arange(-8, 9) * (np.pi / 4), [0.0, 1e-300, -1e-9]])
    sn, cs = C.c_double(), C.c_double()
    got = np.empty((len(x), 2))
    for i, v in enumerate(x):
        hostsim.hs_sincos(float(v), C.byref(sn), C.byref(cs))
        got[i] = sn.value, cs.value
    ref = np.stack([np.sin(x), np.cos(x)], axis=1)
    err = np.abs(got - ref)
    # (absolute floor: at multiples of pi/2 the reduced argument carries the 1e-32 of the 3-piece pi/2)
    assert np.all(err <= 2 * np.spacing(np.maximum(np.abs(ref), 1e-300)) + 1e-30)


# ---- round 4: the ordering keys recomputed in the streaming passes (sl_level.hip) ---------------------
@pytest.mark.parametrize("name,kw,limits", [
    ("1d", dict(num_points=1000), None),
    ("pendulum", dict(num_points=[17, 40], dynamics="linear"), None),
    ("pendulum", dict(num_points=[9, 24], dynamics="linear"), [[-1.0, 1.0], [-0.5, 0.75]]),
    ("cartpole", dict(num_points=[5, 6, 7, 16], dynamics="linear"), None),
    ("cartpole", dict(num_points=8, dynamics="linear"), None),
    ("cartpole", dict(num_points=[3, 4, 5, 24], dynamics="linear"), [[-1, 1], [-2, 2], [-0.5, 0.5], [-4, 4]]),
])
def test_row_values_equal_the_value_pass_bit_for_bit(hostsim, name, kw, limits):
    """SlRowValues (8 cells of a grid row per thread, the prefix of the ordered sums shared) against
    sl_quadratic on the np.linspace points (what k_values writes) and against the oracle's values."""
    case = cases.make_case(name, **kw)
    if limits is not None:
        case["limits"] = [[float(a), float(b)] for a, b in limits]
    grid, desc = _describe(case)
    assert hostsim.hs_values_implicit_ok(C.byref(desc)) == 1
    n = grid.nindex
    rows, points = np.zeros(n), np.zeros(n)
    assert hostsim.hs_row_values(C.byref(desc), C.c_int64(n), rows.ctypes.data_as(C.c_void_p),
                                 points.ctypes.data_as(C.c_void_p)) == 0
    assert_array_equal(rows, points)
    olyap = cases.oracle_lyapunov(case)
    assert_array_equal(rows, olyap.values)
    # a negated quadratic (-V as in the reinforcement-learning notebooks)
    desc.value.negate = 1
    assert hostsim.hs_row_values(C.byref(desc), C.c_int64(n), rows.ctypes.data_as(C.c_void_p),
                                 points.ctypes.data_as(C.c_void_p)) == 0
    assert_array_equal(rows, points)
    assert_array_equal(rows, -olyap.values)


@pytest.mark.parametrize("name,kw,limits,matrix", [
    ("1d", dict(num_points=1000), None, None),
    ("pendulum", dict(num_points=[17, 40], dynamics="linear"), None, None),
    ("pendulum", dict(num_points=[9, 24], dynamics="linear"), [[-1.0, 1.0], [-0.5, 0.75]], None),
    ("cartpole", dict(num_points=[5, 6, 7, 16], dynamics="linear"), None, None),
    ("cartpole", dict(num_points=[3, 4, 5, 24], dynamics="linear"), [[-1, 1], [-2, 2], [-0.5, 0.5], [-4, 4]], None),
    # not symmetric; indefinite; flat along the last axis (ties); a large cross term
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None, [[1.0, 0.7], [-0.2, 0.5]]),
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None, [[1.0, 0.0], [0.0, -0.3]]),
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None, [[1.0, 0.0], [0.0, 0.0]]),
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None, [[1.0, 3.0], [3.0, 1.0]]),
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None, [[1e-9, 0.0], [0.0, 1.0]]),
])
def test_bounded_rows_skip_only_what_is_provably_above_the_level(hostsim, name, kw, limits, matrix):
    """SlRowValues::eight_bounded (the streaming pass of update_safe_set skips rows far above the
    level): whenever it evaluates ONE cell of a group of eight, all eight exact values exceed the
    level, the evaluated cell carries the group's largest (value, index) key and its value is the
    exact one; groups it does not skip are the exact values.  Levels from below the minimum to above
    the maximum, NaN (no failing key) included."""
    case = cases.make_case(name, **kw)
    if limits is not None:
        case["limits"] = [[float(a), float(b)] for a, b in limits]
    if matrix is not None:
        case["P"] = np.array(matrix)
    grid, desc = _describe(case)
    n = grid.nindex
    rows, points = np.zeros(n), np.zeros(n)
    assert hostsim.hs_row_values(C.byref(desc), C.c_int64(n), rows.ctypes.data_as(C.c_void_p),
                                 points.ctypes.data_as(C.c_void_p)) == 0
    exact = rows.reshape(-1, 8)
    hostsim.hs_row_bounded.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    levels = list(np.quantile(rows, [0.0, 0.001, 0.01, 0.2, 0.5, 0.9, 1.0])) + [rows.min() - 1.0, rows.max() + 1.0,
                                                                                float("nan")]
    skipped_any = False
    for vstar in levels:
        top = np.zeros(n // 8, dtype=np.int32)
        vals = np.zeros(n)
        margin = C.c_double(0.0)
        assert hostsim.hs_row_bounded(C.byref(desc), C.c_int64(n), C.c_double(vstar), top.ctypes.data_as(C.c_void_p),
                                      vals.ctypes.data_as(C.c_void_p), C.byref(margin)) == 0
        vals = vals.reshape(-1, 8)
        full = top < 0
        assert_array_equal(vals[full], exact[full])
        skip = ~full
        if np.isnan(vstar) or vstar >= rows.max():
            assert not skip.any()
        if skip.any():
            skipped_any = True
            assert np.all(exact[skip] > vstar)                                   # no cell at or below the level
            # the evaluated cell: exact value, and the lexicographic maximum (value, index) of its group
            picked = top[skip]
            assert set(np.unique(picked)) <= {0, 7}
            got = vals[skip, picked]
            assert_array_equal(got, exact[skip, picked])
            best = np.array([np.flatnonzero(r == r.max())[-1] for r in exact[skip]])
            assert_array_equal(picked, best)
            # strictly monotone groups, or convex ones around the row's minimum (one sign change - to +)
            d = np.diff(exact[skip], axis=1)
            changes = np.count_nonzero(np.diff(np.sign(d), axis=1) != 0, axis=1)
            assert np.all(d != 0) and np.all(changes <= 1)
            assert np.all((changes == 0) | ((d[:, 0] < 0) & (d[:, -1] > 0)))
    if matrix is None or matrix[1][1] != 0.0:
        assert skipped_any                      # the bound is not vacuous
    # the level set's own statistics from the bounded pass equal those of the plain one
    vstar = float(np.quantile(rows, 0.01))
    top = np.zeros(n // 8, dtype=np.int32)
    vals = np.zeros(n)
    hostsim.hs_row_bounded(C.byref(desc), C.c_int64(n), C.c_double(vstar), top.ctypes.data_as(C.c_void_p),
                           vals.ctypes.data_as(C.c_void_p), C.byref(margin))
    vals = vals.reshape(-1, 8)
    seen = np.where((top >= 0)[:, None], -np.inf, vals)          # cells the pass never looked at: -inf
    seen[top >= 0, top[top >= 0]] = vals[top >= 0, top[top >= 0]]
    assert np.count_nonzero(np.where(np.isinf(seen), np.inf, seen) < vstar) == np.count_nonzero(rows < vstar)
    flat = seen.reshape(-1)
    assert flat.max() == rows.max() and np.flatnonzero(flat == flat.max())[-1] == np.flatnonzero(rows == rows.max())[-1]


@pytest.mark.parametrize("name,kw,matrix", [
    ("pendulum", dict(num_points=[12, 32], dynamics="linear"), None),
    ("pendulum", dict(num_points=[9, 128], dynamics="linear"), [[2.0, 0.7], [0.7, 1.0]]),
    ("pendulum", dict(num_points=[7, 120], dynamics="linear"), [[1.0, 3.0], [3.0, 1.0]]),
    ("cartpole", dict(num_points=[4, 5, 6, 64], dynamics="linear"), None),
    ("1d", dict(num_points=256), None),
    # not symmetric; indefinite; flat along the last axis
    ("pendulum", dict(num_points=[12, 64], dynamics="linear"), [[1.0, 0.7], [-0.2, 0.5]]),
    ("pendulum", dict(num_points=[12, 64], dynamics="linear"), [[1.0, 0.0], [0.0, -0.3]]),
    ("pendulum", dict(num_points=[12, 64], dynamics="linear"), [[1.0, 0.0], [0.0, 0.0]]),
])
def test_bounded_spans_of_a_row(hostsim, name, kw, matrix):
    """SlRowValues::span_bounded on the spans the streaming pass gives a thread (the largest divisor
    of the row, at most 16 groups of eight, and smaller ones): a span it clears has every exact value
    above the level, and the one evaluated cell is the exact lexicographic maximum of the span."""
    case = cases.make_case(name, **kw)
    if matrix is not None:
        case["P"] = np.array(matrix)
    grid, desc = _describe(case)
    n = grid.nindex
    rows, points = np.zeros(n), np.zeros(n)
    assert hostsim.hs_row_values(C.byref(desc), C.c_int64(n), rows.ctypes.data_as(C.c_void_p),
                                 points.ctypes.data_as(C.c_void_p)) == 0
    hostsim.hs_span_bounded.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    last = int(grid.num_points[-1])
    spans = sorted({8 * g for g in range(1, 17) if (last // 8) % g == 0})
    assert spans[-1] >= 32
    cleared = 0
    for ncells in spans:
        exact = rows.reshape(-1, ncells)
        for vstar in list(np.quantile(rows, [0.0, 0.01, 0.3, 0.9, 1.0])) + [rows.min() - 1.0, float("nan")]:
            top = np.zeros(n // ncells, dtype=np.int32)
            vtop = np.zeros(n // ncells)
            assert hostsim.hs_span_bounded(C.byref(desc), C.c_int64(n), ncells, C.c_double(vstar),
                                           top.ctypes.data_as(C.c_void_p), vtop.ctypes.data_as(C.c_void_p)) == 0
            skip = top >= 0
            if np.isnan(vstar) or vstar >= rows.max():
                assert not skip.any()
            if not skip.any():
                continue
            cleared += int(skip.sum())
            assert np.all(exact[skip] > vstar)
            assert set(np.unique(top[skip])) <= {0, ncells - 1}
            assert_array_equal(vtop[skip], exact[skip, top[skip]])
            best = np.array([np.flatnonzero(r == r.max())[-1] for r in exact[skip]])
            assert_array_equal(top[skip], best)
    if matrix is None or matrix[1][1] != 0.0:
        assert cleared > 0


def test_bounded_spans_on_random_quadratics(hostsim):
    """The same property on 150 random grids, matrices (definite or not) and levels near the values."""
    rng = np.random.default_rng(5)
    hostsim.hs_span_bounded.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    cleared = 0
    for trial in range(150):
        groups = int(rng.integers(1, 17))
        case = cases.make_case("pendulum", num_points=[int(rng.integers(2, 9)), 8 * groups], dynamics="linear")
        lim = np.sort(rng.normal(size=(2, 2)) * 10.0 ** rng.integers(-3, 3), axis=1)
        case["limits"] = [[float(a), float(b)] for a, b in lim]
        P = rng.normal(size=(2, 2))
        case["P"] = P @ P.T if trial % 3 else P
        grid, desc = _describe(case)
        if not hostsim.hs_values_implicit_ok(C.byref(desc)):
            continue
        n = grid.nindex
        rows, points = np.zeros(n), np.zeros(n)
        assert hostsim.hs_row_values(C.byref(desc), C.c_int64(n), rows.ctypes.data_as(C.c_void_p),
                                     points.ctypes.data_as(C.c_void_p)) == 0
        ncells = 8 * max(g for g in range(1, 17) if groups % g == 0)
        exact = rows.reshape(-1, ncells)
        for vstar in rng.choice(rows, 4):
            for shift in (-1e-12 * abs(vstar), 0.0, 1e-12 * abs(vstar)):
                top = np.zeros(n // ncells, dtype=np.int32)
                vtop = np.zeros(n // ncells)
                assert hostsim.hs_span_bounded(C.byref(desc), C.c_int64(n), ncells, C.c_double(vstar + shift),
                                               top.ctypes.data_as(C.c_void_p), vtop.ctypes.data_as(C.c_void_p)) == 0
                skip = top >= 0
                cleared += int(skip.sum())
                assert np.all(exact[skip] > vstar + shift)
                assert_array_equal(vtop[skip], exact[skip, top[skip]])
                best = np.array([np.flatnonzero(r == r.max())[-1] for r in exact[skip]], dtype=np.int32)
                assert_array_equal(top[skip], best)
    assert cleared > 100


def test_values_stay_explicit_where_the_index_points_differ(hostsim):
    """np.linspace pins the last point of an axis to the upper limit; where index * unit + offset
    rounds to something else (or the last axis is not whole bytes, or V is a table) the passes read V."""
    case = cases.make_case("pendulum", num_points=[17, 40], dynamics="linear")
    case["limits"] = [[-1.0, 1.03], [-0.97, 1.0]]
    grid, desc = _describe(case)
    exact = all((int(n) - 1) * u + o == hi for n, u, o, (_, hi) in
                zip(grid.num_points, grid.unit_maxes, grid.offset, grid.limits))
    assert hostsim.hs_values_implicit_ok(C.byref(desc)) == int(exact) == 0
    _, desc = _describe(cases.make_case("pendulum", num_points=[17, 43], dynamics="linear"))
    assert hostsim.hs_values_implicit_ok(C.byref(desc)) == 0            # 43 cells per row
    from safe_learning_amd import _hip
    _, desc = _describe(cases.make_case("pendulum", num_points=[17, 40], dynamics="linear"))
    assert hostsim.hs_values_implicit_ok(C.byref(desc)) == 1
    desc.value.kind = _hip.V_TRI
    assert hostsim.hs_values_implicit_ok(C.byref(desc)) == 0            # a table V is read


def test_fast_key_map_equals_the_reference_map(hostsim):
    hostsim.hs_vbits_fast.restype = C.c_uint64
    hostsim.hs_vbits_fast.argtypes = [C.c_double]
    rng = np.random.default_rng(0)
    samples = np.concatenate([rng.normal(size=2000) * 10.0 ** rng.integers(-300, 300, 2000),
                              [0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 5e-324, -5e-324,
                               np.finfo(float).max, np.finfo(float).min]])
    for v in samples:
        assert hostsim.hs_vbits_fast(float(v)) == hostsim.hs_vbits(float(v)), v


def test_kernel_family_device_functions(hostsim):
    """``sl_kernel_eval`` / ``sl_kernel_diag`` (what ``k_gp_small``, ``k_gp_sweep`` and ``k_bellman``
    evaluate for a head uploaded with ``sl_gp_set_head_kernel``), compiled for the host, against the
    oracle's statement of gpflow 0.4.0's kernels for the cases of the reference-run fixture."""
    import oracle
    from gp_cases import kernel_case_list, kernel_from_spec
    import safe_learning_amd as sl
    from safe_learning_amd import _hip
    rng = np.random.default_rng(11)
    A, B = rng.uniform(-1.5, 1.5, (29, 3)), rng.uniform(-1.5, 1.5, (17, 3))
    A[3] = B[5]                                        # a coinciding pair: euclid_dist's 1e-12
    hostsim.hs_kernel_matrix.restype = C.c_int
    for spec in kernel_case_list():
        for products in spec["kernels"]:
            factors = kernel_from_spec(products, sl)._factors(3)
            ks = _hip.GpKernel()
            ks.nfactors = len(factors)
            for f, (kind, product, variance, inv_ls) in enumerate(factors):
                ks.factor[f].kind, ks.factor[f].product = int(kind), int(product)
                for q in range(3):
                    ks.factor[f].variance[q], ks.factor[f].inv_lengthscales[q] = variance[q], inv_ls[q]
            out, diag = np.zeros((len(A), len(B))), np.zeros(len(A))
            rc = hostsim.hs_kernel_matrix(C.byref(ks), 3, A.ctypes.data_as(C.c_void_p), len(A),
                                          B.ctypes.data_as(C.c_void_p), len(B),
                                          out.ctypes.data_as(C.c_void_p), diag.ctypes.data_as(C.c_void_p))
            assert rc == 0
            okern = kernel_from_spec(products, oracle)
            assert_allclose(out, okern.K(A, B), rtol=1e-12, atol=1e-16)
            assert_allclose(diag, okern.Kdiag(A), rtol=1e-14)
            # the paired evaluation k_gp_small uses: the same bits as one point at a time
            assert hostsim.hs_kernel_eval2_mismatches(C.byref(ks), 3, A.ctypes.data_as(C.c_void_p), len(A),
                                                      B.ctypes.data_as(C.c_void_p), len(B)) == 0
