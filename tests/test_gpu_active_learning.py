"""The active-learning loop of the notebooks (update_safe_set -> get_safe_sample ->
add_data_point; examples/adaptive_safety_verification.ipynb cells 23-25) through the engine,
step by step against the oracle (needs an MI355X)."""

import warnings

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sl():
    import safe_learning_amd
    return safe_learning_amd


def test_perturb_actions(sl):
    rng = np.random.default_rng(0)
    states, actions = rng.normal(size=(7, 2)), rng.normal(size=(7, 1))
    pert = np.array([[0.], [0.3], [-0.3], [5.0]])
    limits = np.array([[-1., 1.]])
    assert_array_equal(sl.perturb_actions(states, actions, pert, limits),
                       oracle.perturb_actions(states, actions, pert, limits))
    assert_array_equal(sl.perturb_actions(states, actions, pert),
                       oracle.perturb_actions(states, actions, pert))


@pytest.mark.parametrize("name,kw,positive", [
    ("pendulum", dict(num_points=40, n_gp=60, tau_scale=0.0), True),
    ("pendulum", dict(num_points=40, n_gp=60, tau_scale=0.0), False),
    ("cartpole", dict(num_points=7, n_gp=80, tau_scale=0.0, noise_std=0.001, signal_std=0.01), True),
])
def test_active_learning_loop(sl, name, kw, positive):
    from safe_learning_amd.benchmarks import build_lyapunov, _true_dynamics_numpy
    case = cases.make_case(name, **kw)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    perturbations = np.array([[0.], [0.1], [-0.1]])
    limits = np.array([[-1., 1.]])
    for step in range(3):
        lyap.update_safe_set()
        olyap.update_safe_set()
        assert_array_equal(lyap.safe_set, olyap.safe_set)
        assert lyap.c_max == olyap.c_max
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            np.random.seed(step)
            sa, bound = sl.get_safe_sample(lyap, perturbations, limits, positive=positive,
                                           num_samples=200)
            np.random.seed(step)
            osa, obound = oracle.get_safe_sample(olyap, perturbations, limits, positive=positive,
                                                 num_samples=200)
        assert_allclose(bound, obound, rtol=1e-7)
        assert_allclose(sa, osa, rtol=0, atol=0)
        measurement = _true_dynamics_numpy(case, osa)
        lyap.dynamics.add_data_point(sa, measurement)          # rank-one update
        olyap.dynamics.add_data_point(osa, measurement)        # full rebuild (reference)
    lyap.update_safe_set()
    olyap.update_safe_set()
    assert_array_equal(lyap.safe_set, olyap.safe_set)


@pytest.mark.parametrize("name,n0,cfg", [("pendulum", 30, None), ("cartpole", 200, None),
                                           ("cartpole", 508, "2")])
def test_incremental_gp_upload_matches_full_upload(sl, name, n0, cfg, monkeypatch):
    """add_data_point uploads only the new row of L^-1 / alpha / X (sl_gp_append_point); the sweep
    afterwards must be identical, record for record, to the one of a freshly built model with the
    same data - also across the padded capacity of the head (which forces a re-pack)."""
    from safe_learning_amd.benchmarks import build_lyapunov, _true_dynamics_numpy
    from test_gpu_lyapunov import _engine_records
    if cfg is not None:
        monkeypatch.setenv("SL_GP_CFG", cfg)
    npts = 24 if name == "pendulum" else 6
    case = cases.make_case(name, num_points=npts, n_gp=n0, tau_scale=0.0, signal_std=0.03,
                           noise_std=0.001, lengthscale=1.0)
    lyap = build_lyapunov(case)
    _engine_records(lyap)                                  # uploads the head
    uploads = {"full": 0, "rows": 0}
    ctx = lyap._ctx
    full, rows = ctx.gp_set_head, ctx.gp_append_point
    ctx.gp_set_head = lambda *a, **k: (uploads.__setitem__("full", uploads["full"] + 1), full(*a, **k))[1]
    ctx.gp_append_point = lambda *a, **k: (uploads.__setitem__("rows", uploads["rows"] + 1), rows(*a, **k))[1]
    rng = np.random.default_rng(21)
    d = case["d"]
    for step in range(3):
        x_new = rng.uniform(-1, 1, (3, d + 1))
        lyap.dynamics.add_data_point(x_new, _true_dynamics_numpy(case, x_new))
        _, neg, rec = _engine_records(lyap)
        dyn = case["dynamics"]
        fresh = dict(case)
        fresh["dynamics"] = dict(dyn, X=lyap.dynamics.X.copy(), Y=lyap.dynamics.Y.copy())
        ref = build_lyapunov(fresh)
        _, rneg, rrec = _engine_records(ref)
        # rank-one factors vs a fresh Cholesky: equal up to rounding of the factorisation
        assert_allclose(rec, rrec, rtol=1e-8, atol=1e-12)
        assert_array_equal(neg, rneg)
    if cfg is None:
        assert uploads["rows"] == 9 and uploads["full"] == 0
    else:                                                  # 508 + 9 points cross the 512-row panel
        assert uploads["full"] >= 1 and uploads["rows"] >= 4


def test_appended_inputs_are_bit_identical_to_a_fresh_upload(sl):
    """The scaled training inputs of a head extended by sl_gp_append_point equal those of a fresh
    sl_gp_set_head of the same data BIT FOR BIT, also for lengthscales that are not powers of two
    (x / l, not x * (1 / l)): cells next to the decrease threshold must not depend on the upload
    history."""
    from safe_learning_amd.benchmarks import build_lyapunov, _true_dynamics_numpy
    from test_gpu_lyapunov import _engine_records
    case = cases.make_case("pendulum", num_points=24, n_gp=40, tau_scale=0.0, signal_std=0.03,
                           noise_std=0.001, lengthscale=0.37)
    lyap = build_lyapunov(case)
    _engine_records(lyap)
    rng = np.random.default_rng(5)
    x_new = rng.uniform(-1, 1, (7, 3))
    lyap.dynamics.add_data_point(x_new, _true_dynamics_numpy(case, x_new))
    _engine_records(lyap)                                   # appends the seven points
    fresh = dict(case)
    fresh["dynamics"] = dict(case["dynamics"], X=lyap.dynamics.X.copy(), Y=lyap.dynamics.Y.copy())
    ref = build_lyapunov(fresh)
    _engine_records(ref)
    n, p = len(lyap.dynamics.X), 3
    xs, xs_ref = lyap._ctx.gp_inputs(0, n, p), ref._ctx.gp_inputs(0, n, p)
    assert_array_equal(xs, xs_ref)
    assert_array_equal(xs, (lyap.dynamics.X / 0.37).T)
    assert np.any(xs != (lyap.dynamics.X * (1.0 / 0.37)).T)   # the two roundings do differ


def _one_dimensional_gp_case(n_cells=201, n_gp=25):
    """A 1-D grid with a GP model of x+ = 1.25 x + 0.6 u and a policy that saturates at +-0.3 (the
    shape of ``1d_example.ipynb``): the level set stops where the saturated control no longer
    contracts - 139 of 201 cells are safe."""
    rng = np.random.default_rng(3)
    true = np.array([[1.25, 0.6]])
    X = rng.uniform(-1, 1, (n_gp, 2))
    Y = X @ true.T + rng.normal(0, 0.002, (n_gp, 1))
    return dict(name="1d-gp", stack=False, d=1, m=1, limits=[[-1., 1.]], num_points=[n_cells],
                K=np.array([[-0.9]]), saturate=(-0.3, 0.3), P=np.array([[1.]]),
                lv=("abs_linear", 2 * np.array([[1.]])), lf=1.8, tau=0.2 / (n_cells - 1),
                initial_radius=0.1,
                dynamics={"kind": "gp", "X": X, "Y": Y, "variance": 0.02 ** 2,
                          "lengthscales": np.full(2, 1.0), "noise_variance": 0.002 ** 2,
                          "prior": np.array([[1.2, 0.5]]), "beta": 2.0})


@pytest.mark.parametrize("positive", [True, False])
def test_get_safe_sample_over_an_action_grid(sl, positive):
    """``get_safe_sample(lyapunov, perturbations=None, actions=grid)`` (``lyapunov.py:737-741``): every
    safe state with every action of a grid - in the engine the pairs are built on the device (the
    safe states used to travel to the host and back for the meshgrid)."""
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _one_dimensional_gp_case()
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    lyap.update_safe_set()
    olyap.update_safe_set()
    assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert cases.initial_safe_mask(case).sum() + 10 < olyap.safe_set.sum() < 190
    actions = np.linspace(-1, 1, 21)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        pair, bound = sl.get_safe_sample(lyap, None, None, positive=positive, actions=actions)
        opair, obound = oracle.get_safe_sample(olyap, None, None, positive=positive, actions=actions)
    assert_array_equal(pair, opair)
    assert_allclose(bound, obound, rtol=1e-7)


@pytest.mark.parametrize("n,density", [(1, 1.0), (63, 0.5), (64, 0.0), (1000, 0.3), (16384, 1.0), (16385, 0.01),
                                       (70001, 0.5), (3003501, 0.03), (3003501, 1.0)])
def test_safe_indices_are_numpy_where(sl, n, density):
    """``np.where(safe_set)`` of get_safe_sample (``lyapunov.py:729``) from the mask words
    (``sl_bits_count`` + ``sl_bits_to_indices``): every set bit, ascending, bits beyond n ignored."""
    import torch
    from safe_learning_amd import _hip
    rng = np.random.default_rng(n)
    mask = rng.random(n) < density
    ctx = _hip.Context()
    dev = ctx.torch_device
    nwords = (n + 63) // 64
    padded = np.zeros(nwords * 64, dtype=bool)
    padded[:n] = mask
    padded[n:] = True                                   # garbage beyond n must not be reported
    words = torch.from_numpy(np.packbits(padded, bitorder="little").view(np.int64).copy()).to(dev)
    nblocks = -(-nwords // 256)
    counts = torch.empty(nblocks, dtype=torch.int32, device=dev)
    offsets = torch.empty(nblocks + 1, dtype=torch.int64, device=dev)
    total = ctx.bits_count(n, words, counts, offsets)
    want = np.flatnonzero(mask)
    assert total == len(want)
    out = torch.empty(max(total, 1), dtype=torch.int64, device=dev)[:total]
    ctx.bits_to_indices(n, words, offsets, out)
    assert_array_equal(out.cpu().numpy(), want)
