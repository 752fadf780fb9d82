"""The GP posterior - 96 % of the flops of the headline sweep - against the REFERENCE'S OWN GP code.

``tests/golden/reference_gp_posterior.npz`` was computed by the reference's ``GPRCached`` /
``GaussianProcess`` / ``FunctionStack`` (``functions.py:254-307, 357-546``) executed from the
checkout (``tests/golden/make_reference_gp.py``: 14 models, 3 ... 1024 training points, 3 / 5
inputs, ``scale != 1``, cond(K) up to 5e8, zero mean, stacks, added data).  Here, without a GPU:

* the oracle (``oracle/np_functions.py``) reproduces every mean and confidence bound within
  ``reference_gp_tolerance(cond(K))`` (8 eps cond(K): rounding differences times the
  conditioning, see ``tests/gp_cases.py``);
* the HOST side of the engine - ``safe_learning_amd.functions.GPRCached``: explicit inverse
  factor, ``alpha``, no scale (it cancels), rank-one extension for added data - reproduces them too
  when the kernels' contraction ``a = Linv k_x``, ``mean = a . alpha + m(x)``, ``var = s^2 - |a|^2``
  is carried out in NumPy (what ``k_gp_sweep4`` / ``k_gp_small`` do on the matrix cores; the GPU
  test ``tests/test_gpu_reference_gp.py`` compares the kernels themselves with the fixture).
"""

import os

import numpy as np
import pytest

import oracle
from gp_cases import (reference_gp_build_case, reference_gp_case_list, reference_gp_model,
                      reference_gp_tolerance)

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_gp_posterior.npz")
SPECS = reference_gp_case_list()


@pytest.fixture(scope="module")
def fixture():
    return np.load(FIXTURE)


def check(mean, bound, fixture, name, tag, tol):
    want_mean, want_bound = fixture["%s/%s_mean" % (name, tag)], fixture["%s/%s_bound" % (name, tag)]
    scale = np.abs(want_mean).max(axis=0)
    assert np.all(np.abs(mean - want_mean) <= tol * scale), (name, tag, "mean")
    assert np.all(np.abs(bound - want_bound) <= tol * want_bound), (name, tag, "bound")


def test_fixture_is_complete(fixture):
    assert [str(n) for n in fixture["_names"]] == [s["name"] for s in SPECS]
    sizes = {len(fixture[s["name"] + "/X"]) for s in SPECS}
    assert {3, 130, 512, 1024} <= sizes
    assert max(float(fixture[s["name"] + "/cond"]) for s in SPECS) > 1e8


@pytest.mark.parametrize("spec", SPECS, ids=[s["name"] for s in SPECS])
def test_oracle_reproduces_the_reference_posterior(spec, fixture):
    name = spec["name"]
    case = reference_gp_build_case(spec)
    d = case["d"]
    model = reference_gp_model(oracle, spec, case, fixture)
    tol = reference_gp_tolerance(float(fixture[name + "/cond"]))
    assert tol <= 1e-5
    for tag in ("cell", "extra"):
        q = fixture["%s/%s_inputs" % (name, tag)]
        mean, bound = model(q[:, :d], q[:, d:])
        check(mean, bound, fixture, name, tag, tol)
    # the confidence bound is beta * sqrt(var) of build_predict (functions.py:514)
    q = np.vstack((fixture[name + "/cell_inputs"], fixture[name + "/extra_inputs"]))
    _, bound = model(q[:, :d], q[:, d:])
    assert np.allclose(bound, float(fixture[name + "/beta"]) * np.sqrt(fixture[name + "/var"]),
                       rtol=tol, atol=0)


@pytest.mark.parametrize("spec", SPECS, ids=[s["name"] for s in SPECS])
def test_engine_host_factors_reproduce_the_reference_posterior(spec, fixture):
    """What the engine uploads (inverse Cholesky factor, alpha, training inputs) is enough to
    reproduce the reference's posterior: the kernels' contraction in NumPy."""
    import safe_learning_amd.functions as F
    name = spec["name"]
    case = reference_gp_build_case(spec)
    d = case["d"]
    model = reference_gp_model(F, spec, case, fixture)
    heads = model.functions if case["stack"] else [model]
    tol = reference_gp_tolerance(float(fixture[name + "/cond"]))
    for tag in ("cell", "extra"):
        q = fixture["%s/%s_inputs" % (name, tag)]
        means, bounds = [], []
        for h in heads:
            gp = h.gaussian_process
            kx = gp.kern.K(gp.X, q)                                  # [n, m]
            a = gp.cholesky_inverse.dot(kx)
            mean = a.T.dot(gp.alpha)
            if gp.mean_function is not None:
                mean = mean + q.dot(gp.mean_function.matrix.T)
            var = gp.kern.variance - np.sum(a * a, axis=0)
            means.append(mean)
            bounds.append(np.tile((h.beta * np.sqrt(var))[:, None], (1, mean.shape[1])))
        check(np.hstack(means), np.hstack(bounds), fixture, name, tag, tol)
