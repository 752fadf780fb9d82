"""Parity of the dynamic-programming sweep (PolicyIteration) with the oracle (needs an MI355X)."""

import numpy as np
import pytest
import scipy.linalg
from numpy.testing import assert_allclose, assert_array_equal

import cases
import exclusions
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sl():
    import safe_learning_amd
    return safe_learning_amd


ambiguous_points = exclusions.ambiguous_points        # (moved: tests/exclusions.py)


def test_value_iteration_1d_lqr(sl, golden):
    """The system of tests/test_rl.py:29-77: value iteration on a 19-point table with a
    5-vertex piecewise-linear policy; the engine must track the oracle sweep by sweep and end
    within the reference's tolerance of the DLQR solution once the policy is the LQR one."""
    g = golden["policy_iteration_integration"]
    a, b, q, r = (np.array(g[k], dtype=float) for k in "abqr")
    k, p = oracle.dlqr(a, b, q, r)
    vlim, vnum = g["value_grid"]["limits"], g["value_grid"]["num_points"]
    plim, pnum = g["policy_grid"]["limits"], g["policy_grid"]["num_points"]

    vgrid, pgrid = sl.GridWorld(vlim, vnum), sl.GridWorld(plim, pnum)
    vf = sl.Triangulation(vgrid, 0. * vgrid.all_points, project=True)
    policy = sl.Triangulation(pgrid, -k * pgrid.all_points)
    rl = sl.PolicyIteration(policy, sl.LinearSystem((a, b)),
                            sl.QuadraticFunction(-scipy.linalg.block_diag(q, r)), vf)

    ovgrid, opgrid = oracle.GridWorld(vlim, vnum), oracle.GridWorld(plim, pnum)
    ovf = oracle.Triangulation(ovgrid, 0. * ovgrid.all_points, project=True)
    opolicy = oracle.Triangulation(opgrid, -k * opgrid.all_points)
    orl = oracle.PolicyIteration(opolicy, oracle.LinearSystem((a, b)),
                                 oracle.QuadraticFunction(-scipy.linalg.block_diag(q, r)), ovf)
    assert rl.gamma == orl.gamma == 0.98
    for _ in range(25):
        old = ovf.parameters.copy()
        res = rl.value_iteration()
        orl.value_iteration()
        assert_allclose(vf._host_parameters(), ovf.parameters, rtol=1e-12, atol=1e-13)
        assert_allclose(res, np.max(np.abs(ovf.parameters - old)), rtol=1e-10, atol=1e-12)
    assert_allclose(rl.bellmann_error(), orl.bellmann_error(orl.state_space), rtol=1e-9)
    fv = rl.future_values()
    assert_allclose(fv, orl.future_values(orl.state_space), rtol=1e-12, atol=1e-13)


def _rl_pair(sl, case, n_vgrid, cache=True):
    """cache=False: every sweep recomputes its successors (the tests of the uncached kernels)."""
    d = case["d"]
    limits = case["limits"]
    qmat = -scipy.linalg.block_diag(np.eye(d), 0.1 * np.eye(1))
    vgrid, ovgrid = sl.GridWorld(limits, n_vgrid), oracle.GridWorld(limits, n_vgrid)
    rng = np.random.default_rng(4)
    v0 = -rng.random((vgrid.nindex, 1))
    from safe_learning_amd.benchmarks import build_specs
    policy, dynamics, _, _ = build_specs(case)
    opolicy, odynamics, _, _ = cases.oracle_specs(case)
    vf = sl.Triangulation(vgrid, v0, project=True)
    ovf = oracle.Triangulation(ovgrid, v0, project=True)
    rl = sl.PolicyIteration(policy, dynamics, sl.QuadraticFunction(qmat), vf, gamma=0.95)
    orl = oracle.PolicyIteration(opolicy, odynamics, oracle.QuadraticFunction(qmat), ovf, gamma=0.95)
    if not cache:
        rl.successor_cache(0)
    return rl, orl, vf, ovf


@pytest.mark.parametrize("name,kw,nv", [
    ("pendulum", dict(dynamics="linear"), 21),
    ("pendulum", dict(dynamics="analytic"), 21),
    ("pendulum", dict(n_gp=70), 17),
    ("cartpole", dict(dynamics="analytic"), 6),
    ("cartpole", dict(n_gp=90), 5),
])
def test_value_iteration(sl, name, kw, nv):
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    x = orl.state_space
    nxt = orl.dynamics(x, orl.policy(x))
    nxt = nxt[0] if isinstance(nxt, tuple) else nxt
    ok = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_value_iteration[%s-%s]" % (name, nv), ok, "successor",
                      faces=exclusions.on_boundary_face(ovf, nxt))
    for _ in range(3):
        vf.parameters = ovf.parameters.copy()          # same input table for both
        rl.value_iteration()
        orl.value_iteration()
        assert_allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(dynamics="analytic"), 15, 9),
    ("pendulum", dict(n_gp=70), 15, 9),             # matrix-core sweep, 2 column blocks (of 3)
    ("pendulum", dict(n_gp=70), [16, 32], 3),       # 1 column block, tiles aligned with the rows
    ("cartpole", dict(n_gp=90), 5, 9),              # 3 column blocks, ragged tiles
    ("cartpole", dict(n_gp=130), [3, 4, 4, 16], 16),  # 4 column blocks (of 6), n_pad > n
    ("cartpole", dict(n_gp=60, stack=True), 4, 9),  # FunctionStack: one GEMM per head
    ("pendulum", dict(n_gp=70, stack=True), [9, 65], 9),   # the reference's RL example shape
    ("pendulum", dict(n_gp=50, stack=True), 12, 16),
    # last axis = whole wavefronts: the policy-evaluation sweep that follows runs on the matrix
    # cores (random value table => noisy greedy policy: several GEMM rounds and the scalar tail)
    ("pendulum", dict(n_gp=70), [12, 64], 9),
    ("pendulum", dict(n_gp=70), [5, 128], 2),       # at most two distinct actions per row
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 16),
    # rows that are not whole wavefronts: ragged last segment, rows straddling the wavefronts
    ("pendulum", dict(n_gp=70), [9, 65], 5),
    ("pendulum", dict(n_gp=70), [6, 101], 3),
    ("cartpole", dict(n_gp=90), [3, 3, 2, 70], 9),
])
def test_discrete_policy_optimization(sl, name, kw, nv, na):
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    actions = np.linspace(-1, 1, na)[:, None]
    grid, ogrid = vf.discretization, ovf.discretization
    rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
    orl.policy = oracle.Triangulation(ogrid, np.zeros((ogrid.nindex, 1)))
    q = rl.discrete_policy_optimization(actions, return_values=True)
    oq, obest = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok_q = np.ones_like(oq, dtype=bool)
    for a, action in enumerate(actions):
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        nxt = nxt[0] if isinstance(nxt, tuple) else nxt
        ok_q[:, a] = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_discrete_policy_optimization[%s-%s-%d]" % (name, nv, na), ok_q, "successor")
    got_q = q.cpu().numpy()
    assert_allclose(got_q[ok_q], oq[ok_q], rtol=1e-9, atol=1e-12)
    # the greedy action may differ only where an excluded entry or a rounding tie decides
    best = rl.policy._host_parameters()[:, 0]
    obest_val = orl.policy.parameters[:, 0]
    top2 = np.sort(oq, axis=1)[:, -2:]
    tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    assert not np.any((best != obest_val) & ok_q.all(axis=1) & ~tie)
    # one more sweep with the greedy table policy (interpolated at its own vertices like the
    # reference does), from identical inputs
    rl.policy.parameters = orl.policy.parameters.copy()
    vf.parameters = ovf.parameters.copy()
    u = orl.policy(x)
    nxt = orl.dynamics(x, u)
    nxt = nxt[0] if isinstance(nxt, tuple) else nxt
    label = "test_discrete_policy_optimization[%s-%s-%d] greedy policy" % (name, nv, na)
    rl.value_iteration()
    got = vf._host_parameters()
    # vertices where the greedy table has no single value: the answer under one of the admissible
    # policy values (nothing is left out)
    amb = exclusions.check_own_vertices(label, orl, orl.policy, x, got)
    ok = ~amb & ~ambiguous_points(ovf, nxt)
    exclusions.report(label, ok | amb, "successor")
    orl.value_iteration()
    assert ok.sum() > 10
    assert_allclose(got[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,kw,nv", [
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64]),        # 77 % of the successors leave the grid
    ("cartpole", dict(dynamics="analytic"), 6),
    ("pendulum", dict(n_gp=70), [5, 128]),
])
def test_projected_successors_on_boundary_faces(sl, name, kw, nv):
    """``project=True`` (functions.py:1187-1190): a successor state outside the value grid is read
    at its projection onto the boundary face.  The reference clips the query 2 eps inside the
    limits before taking unit-cell coordinates (functions.py:706-711, 1116-1124), so a projected
    point lands just inside the outermost rectangle and - the other coordinates being generic -
    in ONE simplex: oracle and engine must agree there.  No exclusions are allowed in this test."""
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    x = orl.state_space
    seen = 0
    for action in (-1.0, 0.37, 1.0):
        u = np.full((len(x), 1), action)
        nxt = orl.dynamics(x, u)
        nxt = nxt[0] if isinstance(nxt, tuple) else nxt
        faces = exclusions.on_boundary_face(ovf, nxt)
        assert not ambiguous_points(ovf, nxt)[faces].any()
        got = rl.future_values(actions=np.array([[action]]))
        ref = orl.future_values(x, actions=u)
        assert_allclose(got[faces], ref[faces], rtol=1e-9, atol=1e-12)
        seen += int(faces.sum())
        # lower AND upper faces occur
        lim = np.asarray(case["limits"], dtype=float)
        assert (nxt <= lim[:, 0]).any() and (nxt >= lim[:, 1]).any()
    assert seen > 0.1 * 3 * len(x)


def test_future_values_with_lyapunov_penalty(sl):
    """future_values(lyapunov=...) (reinforcement_learning.py:107-112): the decrease bound of a
    Lyapunov object as a soft constraint on the value update."""
    from safe_learning_amd.benchmarks import build_lyapunov
    case = cases.make_case("pendulum", num_points=15, n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, 15)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    x = orl.state_space
    for kwargs in (dict(), dict(actions=np.array([[0.3]]))):
        u = orl.policy(x) if not kwargs else np.broadcast_to(kwargs["actions"], (len(x), 1))
        nxt = orl.dynamics(x, u)[0]
        ok = ~ambiguous_points(ovf, nxt)
        got = rl.future_values(None, lyapunov=lyap, lagrange_multiplier=0.7, **kwargs)
        ref = orl.future_values(x, actions=np.array(u), lyapunov=olyap, lagrange_multiplier=0.7)
        exclusions.report("test_future_values_with_lyapunov_penalty", ok, "successor")
        assert_allclose(got[ok], ref[ok], rtol=1e-8, atol=1e-11)
        plain = rl.future_values(None, **kwargs)
        assert np.max(np.abs(plain - got)) > 1e-6          # the penalty is actually there


def test_discrete_policy_optimization_with_constraint(sl):
    """Safety constraint callback of discrete_policy_optimization (:272-275): actions with
    negative slack are ruled out vertex by vertex."""
    case = cases.make_case("pendulum", num_points=15, n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, 15)
    actions = np.linspace(-1, 1, 7)[:, None]
    grid, ogrid = vf.discretization, ovf.discretization
    rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
    orl.policy = oracle.Triangulation(ogrid, np.zeros((ogrid.nindex, 1)))
    x = orl.state_space

    def constraint(action_array):                       # rules out pushing "outwards"
        return -(action_array[:, 0] * x[:, 0]) + 0.05

    q = rl.discrete_policy_optimization(actions, constraint=constraint,
                                        return_values=True).cpu().numpy()
    oq, obest = orl.discrete_policy_optimization(actions, constraint=constraint)
    assert np.isinf(oq).any() and not np.isinf(oq).all(axis=1).all()
    assert_array_equal(np.isinf(q), np.isinf(oq))
    ok_q = np.ones_like(oq, dtype=bool)
    for a, action in enumerate(actions):
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))[0]
        ok_q[:, a] = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_discrete_policy_optimization_with_constraint", ok_q, "successor")
    finite = np.isfinite(oq) & ok_q
    assert_allclose(q[finite], oq[finite], rtol=1e-9, atol=1e-12)
    best = rl.policy._host_parameters()[:, 0]
    masked = np.where(np.isfinite(oq), oq, -np.inf)
    top2 = np.sort(masked, axis=1)[:, -2:]
    tie = ~np.isfinite(top2[:, 0]) | (np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1]))
    assert not np.any((best != orl.policy.parameters[:, 0]) & ok_q.all(axis=1) & ~tie)


def test_future_values_at_arbitrary_states(sl):
    """future_values / bellmann_error away from the grid vertices."""
    case = cases.make_case("pendulum", num_points=15, n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, 15)
    rng = np.random.default_rng(3)
    lim = np.asarray(case["limits"], dtype=float)
    x = rng.uniform(0.1, 0.9, (300, 2)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    nxt = orl.dynamics(x, orl.policy(x))[0]
    ok = ~ambiguous_points(ovf, nxt) & ~ambiguous_points(ovf, x)
    exclusions.report("test_future_values_at_arbitrary_states", ok, "successor")
    assert_allclose(rl.future_values(x)[ok], orl.future_values(x)[ok], rtol=1e-9, atol=1e-12)
    u = rng.uniform(-1, 1, (300, 1))
    nxt = orl.dynamics(x, u)[0]
    ok2 = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_future_values_at_arbitrary_states (explicit actions)", ok2, "successor")
    assert_allclose(rl.future_values(x, actions=u)[ok2], orl.future_values(x, actions=u)[ok2],
                    rtol=1e-9, atol=1e-12)
    assert_allclose(rl.bellmann_error(x[ok]), orl.bellmann_error(x[ok]), rtol=1e-8)


def test_policy_iteration_loop_keeps_tables_on_device(sl):
    """The canonical loop discrete_policy_optimization() -> value_iteration() back to back without
    reading a host table in between (both leave their result on the GPU): `output_dim` and
    `parameters` of a Triangulation must follow the device table."""
    case = cases.make_case("pendulum", num_points=15, n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, 15)
    actions = np.linspace(-1, 1, 9)[:, None]
    for _ in range(3):
        rl.discrete_policy_optimization(actions)
        assert rl.policy.output_dim == 1
        res = rl.value_iteration()
        assert np.isfinite(res)
        assert vf.output_dim == 1
    # the host views appear on demand and agree with the device tables
    assert rl.policy.parameters.shape == (vf.nindex, 1)
    assert vf.parameters.shape == (vf.nindex, 1)
    assert np.isin(rl.policy.parameters, actions).all()
    assert_array_equal(vf.parameters, vf._device_table.cpu().numpy())
    # point evaluation of the (device-resident) greedy policy
    x = orl.state_space[:7]
    assert rl.policy(x).shape == (7, 1)


def test_future_values_per_vertex_actions(sl):
    """future_values(actions=[nindex, m]) pairs row i with vertex i
    (reinforcement_learning.py:89-104); a single row is a constant action."""
    case = cases.make_case("pendulum", num_points=13, dynamics="analytic")
    rl, orl, vf, ovf = _rl_pair(sl, case, 13)
    x = orl.state_space
    rng = np.random.default_rng(8)
    per_vertex = rng.uniform(-1, 1, (len(x), 1))
    nxt = orl.dynamics(x, per_vertex)
    ok = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_future_values_per_vertex_actions", ok, "successor")
    got = rl.future_values(actions=per_vertex)
    ref = orl.future_values(x, actions=per_vertex)
    assert_allclose(got[ok], ref[ok], rtol=1e-9, atol=1e-12)
    const = np.array([[0.3]])
    nxt = orl.dynamics(x, np.broadcast_to(const, (len(x), 1)))
    ok = ~ambiguous_points(ovf, nxt)
    exclusions.report("test_future_values_per_vertex_actions (constant)", ok, "successor")
    got = rl.future_values(actions=const)
    ref = orl.future_values(x, actions=np.broadcast_to(const, (len(x), 1)))
    assert_allclose(got[ok], ref[ok], rtol=1e-9, atol=1e-12)
    assert_allclose(rl.future_values(actions=np.broadcast_to(const, (len(x), 1)))[ok], ref[ok],
                    rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        rl.future_values(actions=per_vertex[:5])


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(n_gp=70), [12, 64], 9),          # 18 (action, output) rows: 2 row blocks
    ("pendulum", dict(n_gp=70), [5, 128], 2),          # 1 row block, two wavefronts per grid row
    ("pendulum", dict(n_gp=100), [7, 64], 16),         # 32 rows, 100 points padded to 128
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 9),     # 36 rows: the 64^4 x 9 sweep's shape
    ("cartpole", dict(n_gp=130), [2, 3, 2, 128], 12),  # 48 rows, all three row blocks full
])
def test_bellman_sweep_4x4x4_kernel(sl, name, kw, nv, na, monkeypatch):
    """k_bellman4 (v_mfma_f64_4x4x4_4b_f64, sl_bellman4.hip) takes the max sweeps whose last grid
    axis is a multiple of 64 cells: its action values against the oracle's
    (reinforcement_learning.py:266-279) and against k_bellman_mfma's on the same inputs."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None]
    results = {}
    # "1": k_bellman4s (B operand shared by the workgroup) + k_bellman_lookup (shipped); "split":
    # k_bellman4 + k_bellman_lookup; "fused": k_bellman4 with its own epilogue; "0": the 16x16x4 kernel
    for flag in ("1", "split", "fused", "0"):
        monkeypatch.setenv("SL_BELLMAN4", "0" if flag == "0" else "1")
        monkeypatch.setenv("SL_BELLMAN4_SPLIT", "0" if flag == "fused" else "1")
        monkeypatch.setenv("SL_BELLMAN4_SHARED", "1" if flag == "1" else "0")
        rl, orl, vf, ovf = _rl_pair(sl, case, nv)
        q = rl.discrete_policy_optimization(actions, return_values=True)
        results[flag] = (q.cpu().numpy(), rl.policy.parameters[:, 0].copy(), rl._ctx.last_kernel())
    assert "k_bellman4s" in results["1"][2] and "k_bellman4<" in results["split"][2]
    orl.policy = oracle.Triangulation(ovf.discretization, np.zeros((ovf.discretization.nindex, 1)))
    oq, obest = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok = np.ones(len(x), dtype=bool)
    for action in actions:
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        ok &= ~ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
    exclusions.report("test_bellman_sweep_4x4x4_kernel[%s-%s-%d]" % (name, nv, na), ok, "successor")
    q4, best4 = results["1"][:2]
    q16, best16 = results["0"][:2]
    assert_allclose(q4[ok], oq[ok], rtol=1e-9, atol=1e-12)
    assert_allclose(q4, q16, rtol=1e-11, atol=1e-13)
    # the shared-B kernel multiplies (Bt P_j) T_last instead of Bt (P_j T_last): last-bit differences
    assert_allclose(results["split"][0], q4, rtol=1e-12, atol=1e-14)
    assert_array_equal(results["fused"][0], results["split"][0])    # same GEMM, same per-pair arithmetic
    assert_array_equal(results["fused"][1], results["split"][1])
    top2 = np.sort(oq, axis=1)[:, -2:]
    tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    assert not np.any((best4 != actions[obest, 0]) & ok & ~tie)
    assert_array_equal(best4[~tie], best16[~tie])


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(n_gp=70), [12, 64], 9),
    ("cartpole", dict(n_gp=90), 5, 9),
    ("pendulum", dict(dynamics="analytic"), 15, 5),
])
def test_bellman_optimality_sweep(sl, name, kw, nv, na):
    """value_iteration(action_space): V <- max_a [r + gamma V(f(x, a))] in one sweep, greedy policy
    adopted - against the oracle's table of action values (reinforcement_learning.py:266-279)."""
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    actions = np.linspace(-1, 1, na)[:, None]
    orl.policy = oracle.Triangulation(ovf.discretization, np.zeros((ovf.discretization.nindex, 1)))
    old = ovf.parameters.copy()
    oq, obest = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok = np.ones(len(x), dtype=bool)
    for action in actions:
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        ok &= ~ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
    exclusions.report("test_bellman_optimality_sweep[%s-%s-%d]" % (name, nv, na), ok, "successor")
    res = rl.value_iteration(actions)
    got = vf.parameters[:, 0]
    assert_allclose(got[ok], oq.max(axis=1)[ok], rtol=1e-9, atol=1e-12)
    top2 = np.sort(oq, axis=1)[:, -2:]
    tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    best = rl.policy.parameters[:, 0]
    assert not np.any((best != actions[obest, 0]) & ok & ~tie)
    if ok.all():
        assert_allclose(res, np.max(np.abs(oq.max(axis=1) - old[:, 0])), rtol=1e-9)
    # repeated sweeps contract: the residual decays
    r1 = rl.value_iteration(actions)
    r2 = rl.value_iteration(actions)
    assert r2 < r1


@pytest.mark.parametrize("name,kw,nv,na,style", [
    ("pendulum", dict(n_gp=70), [12, 64], 9, "greedy"),        # the loop's own policy
    ("pendulum", dict(n_gp=70), [5, 128], 3, "random"),        # two tiles per row, <= 3 per tile
    ("pendulum", dict(n_gp=100), [7, 64], 9, "random"),        # 9 distinct per tile: three passes
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 16, "random"),  # 16 distinct: four passes of four
    ("cartpole", dict(n_gp=130), [2, 3, 2, 128], 5, "blocks"), # runs of equal actions, 5 = 4 + 1
    ("pendulum", dict(n_gp=70), [6, 64], 1, "random"),         # a constant table
])
def test_policy_evaluation_4x4x4_kernel(sl, name, kw, nv, na, style, monkeypatch):
    """k_bellman4_policy (sl_bellman4.hip): policy evaluation with a piecewise-constant table
    policy on v_mfma_f64_4x4x4_4b_f64 - one quarter-block GEMM per distinct action of a 64-cell
    tile - against the oracle (reinforcement_learning.py:98-104, 135-140) and against
    k_bellman_policy_mfma (SL_BELLMAN4_POLICY=0) on the same inputs."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None] if na > 1 else np.array([[0.3]])
    rng = np.random.default_rng(5)
    results = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SL_BELLMAN4_POLICY", flag)
        rl, orl, vf, ovf = _rl_pair(sl, case, nv, cache=False)
        grid, ogrid = vf.discretization, ovf.discretization
        n = grid.nindex
        if style == "greedy":
            rl.policy = sl.Triangulation(grid, np.zeros((n, 1)))
            rl.discrete_policy_optimization(actions)
            table = rl.policy._host_parameters().copy()
        elif style == "random":
            table = actions[np.random.default_rng(6).integers(0, na, n)]
        else:
            runs = np.repeat(np.random.default_rng(7).integers(0, na, -(-n // 23)), 23)[:n]
            table = actions[runs]
        if style == "greedy":
            # a Triangulation read at its own vertices, like the reference's loop
            rl.policy = sl.Triangulation(grid, table)
            orl.policy = oracle.Triangulation(ogrid, table)
        else:
            # a per-vertex action table (what value_iteration(action_space) adopts): a random table
            # read through the interpolant has more than 16 distinct values (the `%` wrap-around)
            rl.policy = np.ascontiguousarray(table)
            orl.policy = lambda states, _table=table: _table
        v0 = -rng.random((n, 1)) if flag == "1" else results["v0"]
        results.setdefault("v0", v0)
        vf.parameters = v0.copy()
        ovf.parameters = v0.copy()
        rl.value_iteration()
        kernel = rl._ctx.last_kernel()
        results[flag] = (vf._host_parameters().copy(), rl.last_residual, rl.bellmann_error(), kernel)
    new, old = results["1"], results["0"]
    assert "k_bellman4_policy" in new[3] and "k_bellman_policy_mfma" in old[3], (new[3], old[3])
    assert_allclose(new[0], old[0], rtol=1e-11, atol=1e-13)
    assert_allclose(new[1], old[1], rtol=1e-9)
    assert_allclose(new[2], old[2], rtol=1e-9)
    x = orl.state_space
    nxt = orl.dynamics(x, orl.policy(x))
    ok = ~ambiguous_points(ovf, nxt[0])
    label = "test_policy_evaluation_4x4x4_kernel[%s]" % style
    amb = np.zeros(len(x), dtype=bool)
    if style == "greedy":
        amb = exclusions.check_own_vertices(label, orl, orl.policy, x, new[0])
        ok &= ~amb
    orl.value_iteration()
    exclusions.report(label, ok | amb, "successor")
    assert_allclose(new[0][ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)


def test_policy_evaluation_smooth_policy_keeps_the_16x16x4_kernel(sl):
    """More than 32 distinct policy values: k_bellman4_policy declines, k_bellman_policy_mfma (or
    the scalar kernel) evaluates the policy as before."""
    case = cases.make_case("pendulum", num_points=[6, 64], n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, [6, 64])
    grid, ogrid = vf.discretization, ovf.discretization
    table = np.linspace(-1, 1, grid.nindex)[:, None]
    rl.policy = sl.Triangulation(grid, table)
    orl.policy = oracle.Triangulation(ogrid, table)
    rl.value_iteration()
    assert "k_bellman4_policy" not in rl._ctx.last_kernel()
    x = orl.state_space
    ok = ~ambiguous_points(orl.policy, x) & ~ambiguous_points(ovf, orl.dynamics(x, orl.policy(x))[0])
    exclusions.report("test_policy_evaluation_smooth_policy", ok, "successor")     # a smooth table
    orl.value_iteration()
    assert_allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)


def test_4x4x4_kernels_on_sub_ranges(sl):
    """k_bellman4s and k_bellman4_policy group eight grid rows per workgroup step; an index range
    that starts and ends in the middle of a row (what a rank's shard looks like) must give the
    values of the full-range sweep on its slice, bit for bit."""
    import torch
    case = cases.make_case("pendulum", num_points=[7, 128], n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, [7, 128], cache=False)
    n = vf.discretization.nindex
    actions = np.linspace(-1, 1, 9)[:, None]
    table = actions[np.random.default_rng(3).integers(0, 5, n)]
    policy = np.ascontiguousarray(table)

    def sweeps():
        v_max, argmax, q, _ = rl._sweep(rl.policy, actions, want_q=True)
        max_kernel = rl._ctx.last_kernel()
        v_pol, _, _, _ = rl._sweep(policy, None)
        count = rl._hi - rl._lo
        return (v_max[:count].cpu().numpy(), argmax[:count].cpu().numpy(),
                q[:count].cpu().numpy(), v_pol[:count].cpu().numpy(), max_kernel,
                rl._ctx.last_kernel())

    full = sweeps()
    assert "k_bellman4s" in full[4] and "k_bellman4_policy" in full[5], full[4:]
    for lo, hi in ((64, n - 192), (128 + 64, 128 * 3), (0, 64)):
        rl._lo, rl._hi = lo, hi
        part = sweeps()
        assert "k_bellman4s" in part[4] and "k_bellman4_policy" in part[5], part[4:]
        for got, want in zip(part[:4], full[:4]):
            assert_array_equal(got, want[lo:hi])
    torch.cuda.synchronize()


def test_policy_data_are_reused_until_the_policy_changes(sl):
    """k_bellman4_policy keeps what depends on the policy alone (actions, distinct values, tile
    order) between sweeps, keyed by the context's policy token (bumped whenever a policy table or
    description is handed in): a second sweep reuses them, another table is noticed, results equal
    the uncached ones bit for bit."""
    case = cases.make_case("pendulum", num_points=[12, 64], n_gp=70)
    rl, orl, vf, ovf = _rl_pair(sl, case, [12, 64])
    grid = vf.discretization
    actions = np.linspace(-1, 1, 5)[:, None]
    table = actions[np.random.default_rng(9).integers(0, 5, grid.nindex)]
    rl.policy = sl.Triangulation(grid, table)
    v0 = ovf.parameters.copy()

    def sweep():
        vf.parameters = v0.copy()
        rl.value_iteration()
        return vf._host_parameters().copy(), rl._ctx.last_kernel()

    first, k1 = sweep()
    second, k2 = sweep()
    assert "k_bellman4_policy" in k1 and "reused" not in k1
    assert "reused" in k2
    assert_array_equal(first, second)
    edited = table.copy()
    edited[grid.nindex // 2 + 7] = actions[(np.argmax(actions == edited[grid.nindex // 2 + 7]) + 1) % 5]
    rl.policy.parameters = edited
    third, k3 = sweep()
    assert "reused" not in k3
    assert not np.array_equal(third, first)
    rl.policy.parameters = table
    fourth, k4 = sweep()
    assert "reused" not in k4                     # the key is the last policy's, not a history
    assert_array_equal(fourth, first)


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(n_gp=70), [9, 65], 9),          # one whole and one 1-cell tile per row
    ("pendulum", dict(n_gp=70), [6, 101], 5),
    ("pendulum", dict(n_gp=70), 15, 9),               # rows shorter than a tile
    ("cartpole", dict(n_gp=90), [3, 3, 2, 70], 9),
    ("cartpole", dict(n_gp=90), 5, 9),
    ("pendulum", dict(n_gp=70), [5, 120], 9),         # 120 of 128 cells: taken without forcing
])
def test_max_sweep_ragged_rows_on_4x4x4(sl, name, kw, nv, na, monkeypatch):
    """k_bellman4s takes grids whose last axis is NOT a multiple of 64 cells too (ragged last tile
    of a row, writes masked cell by cell): against k_bellman_mfma (SL_BELLMAN4=0) on the same
    inputs and against the oracle, and on index ranges that cut rows."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None]
    results = {}
    monkeypatch.setenv("SL_BELLMAN4_RAGGED", "1")     # also where the rows fill less than 70 %
    for flag in ("1", "0"):
        monkeypatch.setenv("SL_BELLMAN4", flag)
        rl, orl, vf, ovf = _rl_pair(sl, case, nv)
        q = rl.discrete_policy_optimization(actions, return_values=True)
        results[flag] = (q.cpu().numpy(), rl._ctx.last_kernel())
    assert "k_bellman4s" in results["1"][1] and "k_bellman4" not in results["0"][1], \
        (results["1"][1], results["0"][1])
    assert_allclose(results["1"][0], results["0"][0], rtol=1e-11, atol=1e-13)
    orl.policy = oracle.Triangulation(ovf.discretization, np.zeros((ovf.discretization.nindex, 1)))
    oq, _ = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok = np.ones(len(x), dtype=bool)
    for action in actions:
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        ok &= ~ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
    exclusions.report("test_max_sweep_ragged_rows_on_4x4x4[%s-%s-%d]" % (name, nv, na), ok, "successor")
    assert_allclose(results["1"][0][ok], oq[ok], rtol=1e-9, atol=1e-12)
    # a shard-like range (multiples of 64 cells) that cuts rows
    monkeypatch.setenv("SL_BELLMAN4", "1")
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    n = vf.discretization.nindex
    if n >= 256:
        lo, hi = 64, (n // 64 - 1) * 64
        rl._lo, rl._hi = lo, hi
        _, _, q, _ = rl._sweep(rl.policy, actions, want_q=True)
        assert "k_bellman4s" in rl._ctx.last_kernel()
        assert_array_equal(q[:hi - lo].cpu().numpy(), results["1"][0][lo:hi])


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(n_gp=70), [9, 65], 5),
    ("pendulum", dict(n_gp=70), [6, 101], 9),
    ("pendulum", dict(n_gp=70), 15, 3),
    ("cartpole", dict(n_gp=90), [3, 3, 2, 70], 9),
    ("pendulum", dict(n_gp=70), [5, 120], 16),
])
def test_policy_evaluation_ragged_rows_on_4x4x4(sl, name, kw, nv, na, monkeypatch):
    """k_bellman4_policy on grids whose last axis is not a multiple of 64 cells (masked lanes in the
    ragged tile of a row): against k_bellman_policy_mfma on the same inputs, against the oracle, and
    on an index range that cuts rows."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None]
    monkeypatch.setenv("SL_BELLMAN4_RAGGED", "1")
    results = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SL_BELLMAN4_POLICY", flag)
        rl, orl, vf, ovf = _rl_pair(sl, case, nv)
        n = vf.discretization.nindex
        table = actions[np.random.default_rng(6).integers(0, na, n)]
        rl.policy = np.ascontiguousarray(table)
        rl.value_iteration()
        results[flag] = (vf._host_parameters().copy(), rl.last_residual, rl._ctx.last_kernel())
    assert "k_bellman4_policy" in results["1"][2] and "k_bellman4_policy" not in results["0"][2], \
        (results["1"][2], results["0"][2])
    assert_allclose(results["1"][0], results["0"][0], rtol=1e-11, atol=1e-13)
    assert_allclose(results["1"][1], results["0"][1], rtol=1e-9)
    orl.policy = lambda states, _table=table: _table
    x = orl.state_space
    ok = ~ambiguous_points(ovf, orl.dynamics(x, table)[0])
    orl.value_iteration()
    exclusions.report("test_policy_evaluation_ragged_rows_on_4x4x4[%s-%s-%d]" % (name, nv, na), ok, "successor")
    assert_allclose(results["1"][0][ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)
    # a shard-like range that cuts rows: the same values on its slice
    monkeypatch.setenv("SL_BELLMAN4_POLICY", "1")
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    n = vf.discretization.nindex
    if n >= 256:
        lo, hi = 64, (n // 64 - 1) * 64
        rl._lo, rl._hi = lo, hi
        v_part, _, _, _ = rl._sweep(np.ascontiguousarray(table), None)
        assert "k_bellman4_policy" in rl._ctx.last_kernel()
        assert_array_equal(v_part[:hi - lo].cpu().numpy(), results["1"][0][lo:hi, 0])


@pytest.mark.parametrize("name,kw,nv,na", [
    # (oracle side: 26 % / 29 % / 12 % / 16 % of the (vertex, action) pairs are allowed, 45 % ... 72 %
    # of the vertices allow nothing, the greedy policy changes at 76 % ... 87 % of the vertices)
    ("pendulum", dict(n_gp=70, tau_scale=0.0, gp="informed"), [21, 24], 7),
    ("pendulum", dict(dynamics="analytic", tau_scale=0.0), 19, 5),
    ("cartpole", dict(n_gp=90, tau_scale=0.0, gp="tight"), [4, 4, 3, 16], 9),
    ("cartpole", dict(dynamics="analytic", tau_scale=0.0), [5, 4, 5, 8], 9),
])
def test_discrete_policy_optimization_with_a_lyapunov_constraint(sl, name, kw, nv, na):
    """``discrete_policy_optimization(actions, constraint=lyapunov)``: an action is ruled out at a
    vertex where the Lyapunov decrease condition fails under it (``reinforcement_learning.py:266-278``
    with the slack of ``Lyapunov.safety_constraint``, ``lyapunov.py:378-406``).  Engine: one decrease
    sweep per action into bit masks + a masked arg-max kernel, nothing of size [N, A] leaves the
    device.  Oracle: the reference's loop with the slack as a Python callback."""
    from gp_cases import INFORMED, TIGHT
    from safe_learning_amd.benchmarks import build_lyapunov
    kw = dict(kw)
    kw.update({"informed": INFORMED, "tight": TIGHT, None: {}}[kw.pop("gp", None)])
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    actions = np.linspace(-1, 1, na)[:, None]
    grid, ogrid = vf.discretization, ovf.discretization
    rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
    orl.policy = oracle.Triangulation(ogrid, np.zeros((ogrid.nindex, 1)))
    x = orl.state_space
    init = np.zeros(len(x), dtype=bool)
    init[cases.initial_safe_mask(case)] = True

    near = np.zeros(len(x), dtype=bool)      # vertices where rounding may decide the condition

    def slack(action_array):
        own = olyap.policy
        olyap.policy = lambda states: np.asarray(action_array)[:len(states)]
        try:
            decrease, threshold = olyap._decrease_and_threshold(x)
        finally:
            olyap.policy = own
        scale = np.maximum(np.abs(decrease), np.abs(threshold))
        near[:] |= (np.abs(decrease - threshold) <= 1e-9 * np.maximum(scale, 1e-300))[:, 0] & ~init
        holds = np.less(decrease, threshold)[:, 0] | init
        return np.where(holds, 1.0, -1.0)

    q = rl.discrete_policy_optimization(actions, constraint=lyap, return_values=True)
    oq, obest = orl.discrete_policy_optimization(actions, constraint=slack)
    assert np.isinf(oq).any() and not np.isinf(oq).all(), "the constraint must rule something out"
    ok_q = np.ones_like(oq, dtype=bool)
    for a, action in enumerate(actions):
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        nxt = nxt[0] if isinstance(nxt, tuple) else nxt
        ok_q[:, a] = ~ambiguous_points(ovf, nxt)
    got_q = q.cpu().numpy()
    finite = np.isfinite(oq)
    assert_allclose(got_q[ok_q & finite], oq[ok_q & finite], rtol=1e-9, atol=1e-12)
    # the vetoed entries carry -inf like the reference's `values` (:272-275) - away from the cells
    # whose decrease sits on the threshold, where engine and oracle may decide differently
    assert_array_equal(np.isfinite(got_q)[~near], finite[~near])
    best = rl.policy._host_parameters()[:, 0]
    obest_val = orl.policy.parameters[:, 0]
    masked = np.where(finite, oq, -np.inf)
    top2 = np.sort(masked, axis=1)[:, -2:]
    with np.errstate(invalid="ignore"):
        tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    tie |= ~np.isfinite(top2[:, 1])                   # nothing allowed: index 0 on both sides anyway
    differs = best != obest_val
    assert near.mean() < 0.01
    assert not np.any(differs & ok_q.all(axis=1) & ~tie & ~near)
    # the constraint changed the greedy policy somewhere
    rl.discrete_policy_optimization(actions)
    assert (rl.policy._host_parameters()[:, 0] != best).any()
