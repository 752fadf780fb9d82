"""The safe-exploration loop of ``examples/inverted_pendulum.ipynb`` end to end (needs an MI355X).

The notebook's dynamics model - a ``FunctionStack`` of single-output GPs with the kernels
``Linear + Matern32 * Linear``, built WITHOUT observations - goes through its loop: update the
level set, draw the most uncertain safe state-action pair (``get_safe_sample``), measure the true
dynamics there, ``add_data_point``, again.  Engine and oracle run the loop side by side from the
same model; every stage must agree: safe sets, ``c_max``, the sample, its bound, and therefore the
training sets.  (Each piece has its own parity tests; this one checks that they compose.)
"""

import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


def test_safe_exploration_loop_matches_the_oracle():
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import (_true_dynamics_numpy, build_specs, initial_safe_mask,
                                              notebook_kernels)
    from gp_cases import kernel_from_spec
    case = cases.make_case("pendulum", num_points=[49, 40], n_gp=4, stack=True, tau_scale=0.01,
                           noise_std=0.001)
    d, dyn = case["d"], case["dynamics"]
    specs = notebook_kernels(case)
    # the notebook's variances are those of its own model error; give the prior some width here
    for spec in specs:
        spec[0][0][1]["variance"] = [3e-3, 3e-3, 3e-3]
        spec[1][1][1]["variance"] = 3e-3

    def stack(ns):
        heads = []
        for k in range(d):
            gp = ns.GPRCached(np.empty((0, d + 1)), np.empty((0, 1)), kernel_from_spec(specs[k], ns),
                              ns.LinearSystem((dyn["prior"][[k], :],)),
                              likelihood_variance=dyn["noise_variance"])
            heads.append(ns.GaussianProcess(gp, dyn["beta"]))
        return ns.FunctionStack(heads)

    dynamics, odynamics = stack(sl), stack(oracle)
    policy, _, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=initial_safe_mask(case))
    olyap = cases.oracle_lyapunov(case, dynamics=odynamics)
    perturbations = np.linspace(-0.3, 0.3, 5)[:, None]
    limits = np.array([[-1.0, 1.0]])
    sizes = []
    for stage in range(6):
        lyap.update_safe_set()
        olyap.update_safe_set()
        np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
        assert lyap.c_max == olyap.c_max
        sizes.append(int(lyap.safe_set.sum()))
        pair, bound = sl.get_safe_sample(lyap, perturbations, limits, positive=True)
        opair, obound = oracle.get_safe_sample(olyap, perturbations, limits, positive=True)
        np.testing.assert_array_equal(pair, opair)
        np.testing.assert_allclose(bound, obound, rtol=1e-7)
        measurement = _true_dynamics_numpy(case, opair)
        dynamics.add_data_point(pair, measurement)
        odynamics.add_data_point(opair, measurement)
    assert len(dynamics.functions[0].X) == 6
    np.testing.assert_array_equal(dynamics.functions[0].X, odynamics.functions[0].X)
    assert sizes[-1] > sizes[0], sizes                  # the measurements enlarge the safe set
