"""``k_gp_small`` with the kernels of the reference's notebooks against the reference's own GP code
(needs an MI355X).

``tests/golden/reference_gp_kernels.npz``: ``FunctionStack``s of single-output GPs with
``Linear + Matern32 * Linear`` kernels (``examples/inverted_pendulum.ipynb:152-181``) and a case
with every leaf kind, evaluated by the reference's ``GPRCached`` / ``GaussianProcess`` /
``FunctionStack`` behind the stand-ins (``tests/golden/make_reference_gp_kernels.py``).  Here the
per-cell records of the grid sweep and the explicit-point entry are compared with the FIXTURE, a
whole ``update_safe_set`` with the oracle, and the paths that do not take such kernels must say so.
"""

import os

import numpy as np
import pytest

import cases
import oracle
from gp_cases import (kernel_build_case, kernel_case_list, kernel_from_spec, kernel_model,
                      reference_gp_tolerance)
from test_gpu_reference_gp import sweep_records

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_gp_kernels.npz")
SPECS = kernel_case_list()


@pytest.fixture(scope="module")
def fixture():
    return np.load(FIXTURE)


def build(spec, case, fixture):
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs, initial_safe_mask
    dynamics = kernel_model(sl, spec, case, fixture)
    policy, _, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=initial_safe_mask(case))
    return lyap, dynamics


@pytest.mark.parametrize("spec", SPECS, ids=[s["name"] for s in SPECS])
def test_sweep_reproduces_the_reference_posterior(spec, fixture):
    from safe_learning_amd import _evaluate
    name = spec["name"]
    case = kernel_build_case(spec)
    d = case["d"]
    lyap, dynamics = build(spec, case, fixture)
    lyap._upload_model()
    lyap._refresh_init_bits()
    tol = reference_gp_tolerance(float(fixture[name + "/cond"]))
    rec, kernels = sweep_records(lyap, fixture[name + "/cell_index"])
    assert all(k.startswith("k_gp_small") for k in kernels), kernels
    want_mean, want_bound = fixture[name + "/cell_mean"], fixture[name + "/cell_bound"]
    scale = np.abs(want_mean).max(axis=0)
    assert np.all(np.abs(rec[:, 2:2 + d] - want_mean) <= tol * scale), (name, "mean")
    assert np.all(np.abs(rec[:, 2 + d:] - want_bound) <= tol * want_bound), (name, "bound")
    q = fixture[name + "/extra_inputs"]
    mean, bound = _evaluate.dynamics(dynamics, q[:, :d], q[:, d:])
    want_mean, want_bound = fixture[name + "/extra_mean"], fixture[name + "/extra_bound"]
    scale = np.abs(want_mean).max(axis=0)
    assert np.all(np.abs(mean - want_mean) <= tol * scale), (name, "points mean")
    assert np.all(np.abs(bound - want_bound) <= tol * want_bound), (name, "points bound")


@pytest.mark.parametrize("spec", SPECS[:2] + SPECS[3:], ids=[s["name"] for s in SPECS[:2] + SPECS[3:]])
def test_update_safe_set_matches_the_oracle(spec, fixture):
    """The whole level-set rule on a model with notebook kernels: masks, safe sets and c_max equal
    the oracle's, and the workload is not degenerate."""
    name = spec["name"]
    case = kernel_build_case(spec)
    lyap, _ = build(spec, case, fixture)
    odyn = kernel_model(oracle, spec, case, fixture)
    olyap = cases.oracle_lyapunov(case, dynamics=odyn)
    for _ in range(2):
        lyap.update_safe_set()
        olyap.update_safe_set()
        np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
        assert lyap.c_max == olyap.c_max
    n = lyap.discretization.nindex
    neg = np.unpackbits(lyap._d_neg.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
    ref_neg = olyap.negative(olyap.discretization.index_to_state(np.arange(n)))
    assert (neg != ref_neg).sum() == 0
    assert neg.any() and (~neg).any(), name


def test_added_points_follow_the_rank_one_path(fixture):
    """``add_data_point`` on an uploaded model: the engine extends the head in place."""
    spec = SPECS[2]
    case = kernel_build_case(spec)
    lyap, dynamics = build({k: v for k, v in spec.items() if k != "add_points"}, case, fixture)
    lyap.update_safe_set()
    for x, y in zip(fixture[spec["name"] + "/added_x"], fixture[spec["name"] + "/added_y"]):
        dynamics.add_data_point(x[None, :], y[None, :])
    lyap._upload_model()
    lyap._refresh_init_bits()
    d = case["d"]
    tol = reference_gp_tolerance(float(fixture[spec["name"] + "/cond"]))
    rec, _ = sweep_records(lyap, fixture[spec["name"] + "/cell_index"])
    want_mean, want_bound = fixture[spec["name"] + "/cell_mean"], fixture[spec["name"] + "/cell_bound"]
    assert np.all(np.abs(rec[:, 2:2 + d] - want_mean) <= tol * np.abs(want_mean).max(axis=0))
    assert np.all(np.abs(rec[:, 2 + d:] - want_bound) <= tol * want_bound)


def _matern_stack(ns, kern_lib, case, n):
    dyn = case["dynamics"]
    heads = []
    for k in range(case["d"]):
        kern = kernel_from_spec([[("matern32", dict(input_dim=3, variance=0.05 ** 2, lengthscales=[0.5, 0.6, 0.7], ARD=True))],
                                 [("linear", dict(input_dim=3, variance=[1e-4, 2e-4, 1e-4], ARD=True))]], kern_lib)
        gp = ns.GPRCached(dyn["X"][:n], dyn["Y"][:n, [k]], kern, ns.LinearSystem((dyn["prior"][[k], :],)),
                          likelihood_variance=dyn["noise_variance"])
        heads.append(ns.GaussianProcess(gp, dyn["beta"]))
    return ns.FunctionStack(heads)


@pytest.mark.parametrize("n,cfg", [(300, None), (130, "3"), (130, "1")], ids=["n300", "n130_cfg3", "n130_cfg1"])
def test_large_sets_run_on_the_16x16x4_kernel(n, cfg, monkeypatch):
    """More than 256 training points (or a forced configuration): ``k_gp_sweep`` evaluates the
    leaves; records, masks and the safe set equal the oracle's."""
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs, initial_safe_mask
    if cfg is not None:
        monkeypatch.setenv("SL_GP_CFG", cfg)
        monkeypatch.setenv("SL_GP_SMALL", "0")
    case = cases.make_case("pendulum", num_points=[40, 33], n_gp=300, stack=True, tau_scale=0.01,
                           noise_std=0.001)
    dynamics, odynamics = _matern_stack(sl, sl, case, n), _matern_stack(oracle, oracle, case, n)
    policy, _, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=initial_safe_mask(case))
    olyap = cases.oracle_lyapunov(case, dynamics=odynamics)
    lyap.update_safe_set()
    olyap.update_safe_set()
    assert lyap._ctx.last_kernel().startswith("k_gp_sweep<"), lyap._ctx.last_kernel()
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max
    nidx = lyap.discretization.nindex
    got, _ = sweep_records(lyap, np.arange(nidx))
    rec = cases.oracle_cell_records(olyap, np.arange(nidx))
    np.testing.assert_allclose(got[:, 2:], rec[:, 2:], rtol=1e-8, atol=1e-12)
    neg = got[:, 0] < got[:, 1]
    assert neg.any() and (~neg).any()


@pytest.mark.parametrize("nv,na", [(15, 9), ([9, 65], 5)], ids=["15x15", "9x65"])
def test_policy_iteration_with_notebook_kernels(nv, na):
    """``inverted_pendulum.ipynb`` hands its GP ``FunctionStack`` to ``PolicyIteration``: value
    iteration and the discrete policy optimisation on a model with the notebook's kernels equal
    the oracle's (the matrix-core sweeps generate RBF values; such models take ``k_bellman``)."""
    import scipy.linalg
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs, notebook_kernels
    from test_gpu_rl import ambiguous_points
    case = cases.make_case("pendulum", num_points=nv, n_gp=70, stack=True)
    case["dynamics"]["kernels"] = notebook_kernels(case)
    policy, dynamics, _, _ = build_specs(case)
    opolicy, odynamics, _, _ = cases.oracle_specs(case)
    qmat = -scipy.linalg.block_diag(np.eye(2), 0.1 * np.eye(1))
    vgrid, ovgrid = sl.GridWorld(case["limits"], nv), oracle.GridWorld(case["limits"], nv)
    v0 = -np.random.default_rng(4).random((vgrid.nindex, 1))
    vf, ovf = sl.Triangulation(vgrid, v0, project=True), oracle.Triangulation(ovgrid, v0, project=True)
    rl = sl.PolicyIteration(policy, dynamics, sl.QuadraticFunction(qmat), vf, gamma=0.95)
    orl = oracle.PolicyIteration(opolicy, odynamics, oracle.QuadraticFunction(qmat), ovf, gamma=0.95)
    x = orl.state_space
    ok = ~ambiguous_points(ovf, orl.dynamics(x, orl.policy(x))[0])
    assert ok.mean() > 0.9
    for _ in range(2):
        vf.parameters = ovf.parameters.copy()
        rl.value_iteration()
        orl.value_iteration()
        assert rl._ctx.last_kernel().startswith("k_bellman<"), rl._ctx.last_kernel()
        np.testing.assert_allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)
    actions = np.linspace(-1, 1, na)[:, None]
    rl.policy = sl.Triangulation(vgrid, np.zeros((vgrid.nindex, 1)))
    orl.policy = oracle.Triangulation(ovgrid, np.zeros((ovgrid.nindex, 1)))
    vf.parameters = ovf.parameters.copy()
    q = rl.discrete_policy_optimization(actions, return_values=True).cpu().numpy()
    oq, _ = orl.discrete_policy_optimization(actions)
    ok_q = np.ones_like(oq, dtype=bool)
    for a, action in enumerate(actions):
        ok_q[:, a] = ~ambiguous_points(ovf, orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))[0])
    assert ok_q.mean() > 0.9
    np.testing.assert_allclose(q[ok_q], oq[ok_q], rtol=1e-9, atol=1e-12)


def test_notebook_flow_from_a_model_without_observations(fixture):
    """``inverted_pendulum.ipynb:166-176`` builds its GPs on ``np.empty((0, 3))`` and adds the
    observations one by one: the engine's posterior is the prior first, follows every
    ``add_data_point`` (full upload for the first point, rank-one rows afterwards), and the level
    sets equal the oracle's at every stage."""
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs, initial_safe_mask
    from gp_cases import kernel_from_spec
    spec = SPECS[0]
    case = kernel_build_case(spec)
    d, dyn = case["d"], case["dynamics"]

    def stack(ns):
        heads = []
        for k in range(d):
            kern = kernel_from_spec(spec["kernels"][k], ns)
            gp = ns.GPRCached(np.empty((0, d + 1)), np.empty((0, 1)), kern,
                              ns.LinearSystem((dyn["prior"][[k], :],)),
                              likelihood_variance=dyn["noise_variance"])
            heads.append(ns.GaussianProcess(gp, dyn["beta"]))
        return ns.FunctionStack(heads)

    dynamics, odynamics = stack(sl), stack(oracle)
    policy, _, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=initial_safe_mask(case))
    olyap = cases.oracle_lyapunov(case, dynamics=odynamics)
    n = lyap.discretization.nindex
    sizes = []
    for stage in range(4):
        if stage:
            for x, y in zip(dyn["X"][3 * stage - 3:3 * stage], dyn["Y"][3 * stage - 3:3 * stage]):
                dynamics.add_data_point(x[None, :], y[None, :])
                odynamics.add_data_point(x[None, :], y[None, :])
        lyap.update_safe_set()
        olyap.update_safe_set()
        assert lyap._ctx.last_kernel().startswith("k_gp_small")
        np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
        assert lyap.c_max == olyap.c_max
        rec = cases.oracle_cell_records(olyap, np.arange(n))
        got, _ = sweep_records(lyap, np.arange(n))
        np.testing.assert_allclose(got[:, 2:], rec[:, 2:], rtol=1e-9, atol=1e-13)
        sizes.append(int(lyap.safe_set.sum()))
    assert len(dynamics.functions[0].X) == 9
    assert sizes[-1] > sizes[0]                 # the observations shrink the error bounds
