"""The successor cache of the Bellman sweeps (csrc/sl_succ.hip; needs an MI355X).

``reinforcement_learning.py:89-104``: ``dynamics(states, actions)`` never sees the value table, which
enters at ``:101`` only.  The engine therefore locates every (vertex, action) successor once and
serves the later sweeps of a value-iteration loop from that cache.  The contract tested here: the
tables, greedy policies and residuals of cached sweeps are IDENTICAL (bit for bit) to the
recomputing kernels', whatever invalidates a successor (new GP data, other hyper-parameters, another
action set, another range) is noticed, and policy evaluation with a greedy table selects the cached
entry of each vertex's action."""

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases
import exclusions
import oracle
from test_gpu_rl import _rl_pair, ambiguous_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sl():
    import safe_learning_amd
    return safe_learning_amd


def _loop(rl, vf, actions, sweeps, v0):
    """`sweeps` Bellman optimality backups from v0: tables, greedy tables, residuals, kernels."""
    vf.parameters = v0.copy()
    out = []
    for _ in range(sweeps):
        res = rl.value_iteration(actions)
        out.append((vf._host_parameters().copy(), rl.policy._host_parameters().copy(), res,
                    rl._ctx.last_kernel()))
    return out


@pytest.mark.parametrize("name,kw,nv,na,fill_kernel", [
    ("pendulum", dict(n_gp=70), [12, 64], 9, "k_bellman_lookup"),          # k_bellman4s + lookup
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 9, "k_bellman_lookup"),     # the 64^4 x 9 shape
    ("pendulum", dict(n_gp=70), [9, 65], 5, "k_bellman_mfma"),             # ragged rows: 16x16x4
    ("pendulum", dict(n_gp=60, stack=True), [9, 65], 9, "k_bellman_mfma"), # FunctionStack heads
    ("pendulum", dict(dynamics="analytic"), 21, 7, "k_bellman<"),          # Euler pendulum
    ("cartpole", dict(dynamics="analytic"), 6, 3, "k_bellman<"),           # Euler cart-pole
    ("pendulum", dict(dynamics="linear"), [16, 33], 16, "k_bellman<"),     # linear system
])
def test_cached_max_sweeps_are_bit_identical(sl, name, kw, nv, na, fill_kernel):
    """20 sweeps of value_iteration(action_space) with the cache against 20 recomputing sweeps from
    the same table: value tables, greedy tables and residuals equal bit for bit; the first sweep
    fills (through the named kernel), the other 19 are served from the cache."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None]
    rl, _, vf, ovf = _rl_pair(sl, case, nv)
    rl_u, _, vf_u, _ = _rl_pair(sl, case, nv, cache=False)
    v0 = ovf.parameters.copy()
    cached = _loop(rl, vf, actions, 20, v0)
    plain = _loop(rl_u, vf_u, actions, 20, v0)
    assert fill_kernel in cached[0][3] and "k_bellman_cached" not in cached[0][3], cached[0][3]
    for sweep, (c, u) in enumerate(zip(cached, plain)):
        assert_array_equal(c[0], u[0], err_msg="value table, sweep %d" % sweep)
        assert_array_equal(c[1], u[1], err_msg="greedy policy, sweep %d" % sweep)
        assert c[2] == u[2], (sweep, c[2], u[2])
        assert "k_bellman_cached" not in u[3]
        if sweep:
            assert "k_bellman_cached" in c[3] and "max" in c[3], c[3]
    info = rl.successor_cache_info
    assert info["valid"] == 1 and info["fills"] == 1 and info["hits"] == 19, info
    assert info["n_actions"] == na and info["bytes"] > 0
    assert rl_u.successor_cache_info["bytes"] == 0 and rl_u.successor_cache_info["max_bytes"] == 0
    # the [N, A] table of action values from the cache
    q_c = rl.discrete_policy_optimization(actions, return_values=True).cpu().numpy()
    q_u = rl_u.discrete_policy_optimization(actions, return_values=True).cpu().numpy()
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    assert_array_equal(q_c, q_u)


def test_cached_sweeps_against_the_oracle(sl):
    """The cached loop tracks the oracle's discrete_policy_optimization + value_iteration
    (reinforcement_learning.py:135-140, 266-279) sweep by sweep."""
    nv, na = [12, 64], 5
    case = cases.make_case("pendulum", num_points=nv, n_gp=70)
    actions = np.linspace(-1, 1, na)[:, None]
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    orl.policy = oracle.Triangulation(ovf.discretization, np.zeros((ovf.discretization.nindex, 1)))
    x = orl.state_space
    ok = np.ones(len(x), dtype=bool)
    for action in actions:
        ok &= ~ambiguous_points(ovf, orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))[0])
    exclusions.report("test_cached_sweeps_against_the_oracle", ok, "successor")
    for sweep in range(4):
        vf.parameters = ovf.parameters.copy()                  # identical inputs every sweep
        oq, _ = orl.discrete_policy_optimization(actions)
        rl.value_iteration(actions)
        assert ("k_bellman_cached" in rl._ctx.last_kernel()) == (sweep > 0)
        assert_allclose(vf.parameters[ok, 0], oq.max(axis=1)[ok], rtol=1e-9, atol=1e-12)
        ovf.parameters = oq.max(axis=1)[:, None]


def test_whatever_moves_a_successor_drops_the_cache(sl):
    """add_data_point, other GP hyper-parameters, another action set, another range: the next sweep
    recomputes (and refills); its results equal a context that never cached."""
    nv, na = [12, 64], 5
    case = cases.make_case("pendulum", num_points=nv, n_gp=70)
    actions = np.linspace(-1, 1, na)[:, None]
    rl, _, vf, ovf = _rl_pair(sl, case, nv)
    rl_u, _, vf_u, _ = _rl_pair(sl, case, nv, cache=False)
    v0 = ovf.parameters.copy()

    def both(acts):
        vf.parameters = v0.copy()
        vf_u.parameters = v0.copy()
        rl.value_iteration(acts)
        kernel = rl._ctx.last_kernel()
        rl_u.value_iteration(acts)
        assert_array_equal(vf._host_parameters(), vf_u._host_parameters())
        return "k_bellman_cached" in kernel

    assert not both(actions)
    assert both(actions)
    # one more observation (functions.py:525-546): the posterior mean moves
    rng = np.random.default_rng(11)
    point = rng.uniform(-0.5, 0.5, (1, 3))
    y = rng.normal(0, 0.1, (1, 2))
    for r in (rl, rl_u):
        r.dynamics.add_data_point(point, y)
    assert not both(actions)
    assert both(actions)
    # another action set of the same size, then a different size
    other = actions * 0.5
    assert not both(other)
    assert both(other)
    assert not both(np.linspace(-1, 1, 3)[:, None])
    assert both(np.linspace(-1, 1, 3)[:, None])
    # new hyper-parameters: a new GP object on the same data
    from safe_learning_amd.benchmarks import build_specs
    case2 = cases.make_case("pendulum", num_points=nv, n_gp=70)
    case2["dynamics"]["lengthscales"] = 1.3 * case2["dynamics"]["lengthscales"]
    _, dyn2, _, _ = build_specs(case2)
    _, dyn2_u, _, _ = build_specs(case2)
    rl.dynamics, rl_u.dynamics = dyn2, dyn2_u
    assert not both(np.linspace(-1, 1, 3)[:, None])
    assert both(np.linspace(-1, 1, 3)[:, None])
    # another reward / gamma / value table: nothing a successor depends on
    import scipy.linalg
    for r in (rl, rl_u):
        r.reward_function = sl.QuadraticFunction(-scipy.linalg.block_diag(2 * np.eye(2), 0.3 * np.eye(1)))
        r.gamma = 0.9
    v0 = -np.random.default_rng(8).random(v0.shape)
    assert both(np.linspace(-1, 1, 3)[:, None])
    # a shard-like range: its own cache, the values of the full sweep on the slice
    n = vf.discretization.nindex
    full = vf._host_parameters().copy()
    rl._lo, rl._hi = 64, n - 128
    vf.parameters = v0.copy()
    v_part, _, _, _ = rl._sweep(rl.policy, np.linspace(-1, 1, 3)[:, None])
    assert "k_bellman_cached" not in rl._ctx.last_kernel()
    v_again, _, _, _ = rl._sweep(rl.policy, np.linspace(-1, 1, 3)[:, None])
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    count = rl._hi - rl._lo
    assert_array_equal(v_part[:count].cpu().numpy(), full[64:n - 128, 0])
    assert_array_equal(v_again[:count].cpu().numpy(), full[64:n - 128, 0])


def test_budget_and_switch(sl, monkeypatch):
    """A cache that does not fit its budget is not built (the sweeps recompute); SL_SUCC_CACHE=0 is
    read when the context is created."""
    nv, na = [12, 64], 5
    case = cases.make_case("pendulum", num_points=nv, n_gp=70)
    actions = np.linspace(-1, 1, na)[:, None]
    rl, _, vf, ovf = _rl_pair(sl, case, nv)
    rl.successor_cache(1000)
    for _ in range(2):
        rl.value_iteration(actions)
        assert "k_bellman_cached" not in rl._ctx.last_kernel()
    assert rl.successor_cache_info["bytes"] == 0
    n = vf.discretization.nindex
    rl.successor_cache(1024 + (na + 1) * n * (8 * 2 + 5) + 64)        # exactly enough
    rl.value_iteration(actions)
    rl.value_iteration(actions)
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    assert 0 < rl.successor_cache_info["bytes"] <= 1024 + (na + 1) * n * 21 + 64
    rl.successor_cache(0)                                               # frees it
    assert rl.successor_cache_info["bytes"] == 0
    rl.value_iteration(actions)
    assert "k_bellman_cached" not in rl._ctx.last_kernel()
    rl.successor_cache(-1)
    rl.value_iteration(actions)
    rl.value_iteration(actions)
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    monkeypatch.setenv("SL_SUCC_CACHE", "0")
    rl2, _, _, _ = _rl_pair(sl, case, nv)
    rl2.value_iteration(actions)
    rl2.value_iteration(actions)
    assert "k_bellman_cached" not in rl2._ctx.last_kernel()


@pytest.mark.parametrize("name,kw,nv,na", [
    ("pendulum", dict(n_gp=70), [12, 64], 9),
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 9),
    ("pendulum", dict(dynamics="analytic"), 21, 7),
])
def test_policy_evaluation_selects_cached_entries(sl, name, kw, nv, na):
    """value_iteration() / bellmann_error() with the greedy table of the loop: every vertex's policy
    value is one of the cached actions, the sweep reads that action's entry (k_bellman_cached,
    policy flavour).  Against the recomputing policy kernels (whose GEMM sees the policy value
    rounded to 2^-40, sl_bellman4.hip: agreement to rounding, not bit for bit) and the oracle."""
    case = cases.make_case(name, num_points=nv, **kw)
    actions = np.linspace(-1, 1, na)[:, None]
    rl, orl, vf, ovf = _rl_pair(sl, case, nv)
    rl_u, _, vf_u, _ = _rl_pair(sl, case, nv, cache=False)
    grid, ogrid = vf.discretization, ovf.discretization
    for r in (rl, rl_u):
        r.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
        r.discrete_policy_optimization(actions)
    assert_array_equal(rl.policy._host_parameters(), rl_u.policy._host_parameters())
    table = rl.policy._host_parameters().copy()
    orl.policy = oracle.Triangulation(ogrid, table)
    x = orl.state_space
    nxt = orl.dynamics(x, orl.policy(x))
    nxt = nxt[0] if isinstance(nxt, tuple) else nxt
    label = "test_policy_evaluation_selects_cached_entries[%s]" % name
    for sweep in range(3):
        v_in = ovf.parameters.copy()
        vf.parameters = v_in.copy()
        vf_u.parameters = v_in.copy()
        res = rl.value_iteration()
        kernel = rl._ctx.last_kernel()
        assert "k_bellman_cached" in kernel and "policy" in kernel, kernel
        res_u = rl_u.value_iteration()
        assert "k_bellman_cached" not in rl_u._ctx.last_kernel()
        got, got_u = vf._host_parameters(), vf_u._host_parameters()
        assert_allclose(got, got_u, rtol=1e-11, atol=1e-13)
        assert_allclose(res, res_u, rtol=1e-9)
        amb = exclusions.check_own_vertices(label, orl, orl.policy, x, got)
        ok = ~amb & ~ambiguous_points(ovf, nxt)
        exclusions.report(label, ok | amb, "successor")
        orl.value_iteration()
        assert_allclose(got[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)
    assert_allclose(rl.bellmann_error(), rl_u.bellmann_error(), rtol=1e-9)
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    assert rl.successor_cache_info["policy_hits"] == 4
    # a per-vertex action table (future_values(actions=...), :89-104) is matched the same way
    rows = actions[np.random.default_rng(2).integers(0, na, grid.nindex)]
    fv = rl.future_values(actions=rows)
    assert "k_bellman_cached" in rl._ctx.last_kernel()
    fv_u = rl_u.future_values(actions=rows)
    assert_allclose(fv, fv_u, rtol=1e-11, atol=1e-13)
    # a policy with a value outside the action set keeps the recomputing kernels
    smooth = np.linspace(-0.9, 0.9, grid.nindex)[:, None]
    rl.policy = sl.Triangulation(grid, smooth)
    rl_u.policy = sl.Triangulation(grid, smooth)
    vf.parameters = v_in.copy()
    vf_u.parameters = v_in.copy()
    rl.value_iteration()
    assert "k_bellman_cached" not in rl._ctx.last_kernel()
    rl_u.value_iteration()
    assert_array_equal(vf._host_parameters(), vf_u._host_parameters())
