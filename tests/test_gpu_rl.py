"""Parity of the dynamic-programming sweep (PolicyIteration) with the oracle (needs an MI355X)."""

import numpy as np
import pytest
import scipy.linalg
from numpy.testing import assert_allclose, assert_array_equal

import cases
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sl():
    import safe_learning_amd
    return safe_learning_amd


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
        assert_allclose(res, np.max(np.abs(ovf.parameters - old)), rtol=1e-10)
    assert_allclose(rl.bellmann_error(), orl.bellmann_error(orl.state_space), rtol=1e-9)
    fv = rl.future_values()
    assert_allclose(fv, orl.future_values(orl.state_space), rtol=1e-12, atol=1e-13)


def _rl_pair(sl, case, n_vgrid):
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
    for _ in range(3):
        rl.value_iteration()
        orl.value_iteration()
        assert_allclose(vf._host_parameters(), ovf.parameters, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,kw,nv", [
    ("pendulum", dict(dynamics="analytic"), 15),
    ("pendulum", dict(n_gp=70), 15),
    ("cartpole", dict(n_gp=90), 5),
    ("cartpole", dict(n_gp=60, stack=True), 4),
])
def test_discrete_policy_optimization(sl, name, kw, nv):
    case = cases.make_case(name, num_points=nv, **kw)
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    actions = np.linspace(-1, 1, 9)[:, None]
    grid, ogrid = vf.discretization, ovf.discretization
    rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
    orl.policy = oracle.Triangulation(ogrid, np.zeros((ogrid.nindex, 1)))
    q = rl.discrete_policy_optimization(actions)
    oq, obest = orl.discrete_policy_optimization(actions)
    assert_allclose(q.cpu().numpy(), oq, rtol=1e-9, atol=1e-12)
    # arg-max may legitimately differ only where two actions tie to rounding
    best = rl.policy._host_parameters()
    differs = best[:, 0] != orl.policy.parameters[:, 0]
    top2 = np.sort(oq, axis=1)[:, -2:]
    assert not np.any(differs & (np.abs(top2[:, 1] - top2[:, 0]) > 1e-9 * np.abs(top2[:, 1])))
    # greedy policy as a table: one more sweep with it
    rl.value_iteration()
    orl.value_iteration()
    ok = ~differs
    assert_allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)
