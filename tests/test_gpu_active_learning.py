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
