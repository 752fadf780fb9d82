"""The oracle's ``Lyapunov`` against safe sets computed by the reference's own ``lyapunov.py``.

``tests/golden/reference_safe_sets.npz`` was produced in the build container by
``tests/golden/make_reference_safe_sets.py``: the reference's ``Lyapunov.update_safe_set`` /
``update_values`` / ``threshold`` / ``v_decrease_bound`` (``lyapunov.py:176-606``),
``get_safe_sample`` (``:609-797``) and ``smallest_boundary_value`` (``:22-56``) executed unmodified
on the reference's own ``functions.py`` / ``examples/utilities.py`` objects (policy, dynamics, V,
L_v; a GP model is the reference's ``GaussianProcess(GPRCached)`` on ``tests/golden/numpy_gpflow.py``
since round 4), with ``tests/golden/numpy_tf.py`` answering the TensorFlow ops they request.  Here every scenario (parameters stored in the fixture)
is replayed on ``oracle.Lyapunov`` through the same step driver: safe set, ``c_max``, refinement
array after every ``update_safe_set`` call, the sample and its bound after every
``get_safe_sample`` call, the value table and the boundary minimum - bit for bit (the bound of a
sample under GP dynamics, a posterior standard deviation, to 1e-10).
"""

import importlib.util
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal

import oracle
from oracle import np_lyapunov

import cases
from conftest import GOLDEN_DIR


def _generator():
    spec = importlib.util.spec_from_file_location(
        "make_reference_safe_sets", os.path.join(GOLDEN_DIR, "make_reference_safe_sets.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


GENERATOR = _generator()
FIXTURE = np.load(os.path.join(GOLDEN_DIR, "reference_safe_sets.npz"))
INDEX = json.loads(str(FIXTURE["_index"]))


def _scenario(entry):
    meta = GENERATOR.from_jsonable(entry["meta"], FIXTURE)
    meta["steps"] = [tuple(step) for step in meta["steps"]]
    meta["case"] = GENERATOR.from_jsonable(entry["case"], FIXTURE)
    return meta


@pytest.fixture
def batch_size():
    saved = np_lyapunov.config.gp_batch_size
    yield
    np_lyapunov.config.gp_batch_size = saved


def test_fixture_covers_the_scenarios_of_the_generator():
    names = [entry["meta"]["name"] for entry in INDEX]
    assert names == [s["name"] for s in GENERATOR.scenarios()]
    assert len(names) >= 19


@pytest.mark.parametrize("entry", INDEX, ids=[entry["meta"]["name"] for entry in INDEX])
def test_safe_sets_equal_the_reference_run(entry, batch_size):
    scenario = _scenario(entry)
    name, case = scenario["name"], scenario["case"]
    np_lyapunov.config.gp_batch_size = scenario["batch"]            # configuration.py:19
    grid = oracle.GridWorld(case["limits"], case["num_points"])
    policy, dynamics, value, lv = cases.oracle_specs(case)
    initial = None if scenario.get("no_initial_set") else cases.initial_safe_mask(case)
    lyap = oracle.Lyapunov(grid, value, dynamics, case["lf"], lv, case["tau"], policy,
                           initial_set=initial)
    lyap.adaptive = bool(scenario.get("adaptive"))
    rtol = scenario.get("values_rtol")              # a network V: BLAS in the oracle
    if rtol:
        assert_allclose(lyap.values, FIXTURE[name + "/values"], rtol=rtol, atol=0)
        assert_allclose(oracle.smallest_boundary_value(value, grid),
                        float(FIXTURE[name + "/boundary"]), rtol=rtol)
    else:
        assert_equal(lyap.values, FIXTURE[name + "/values"])
        assert oracle.smallest_boundary_value(value, grid) == float(FIXTURE[name + "/boundary"])
    records = GENERATOR.replay(scenario, lyap, dynamics, oracle.get_safe_sample,
                               lambda obj: obj.c_max)
    assert len(records) == entry["records"]
    for k, record in enumerate(records):
        for key, got in record.items():
            want = FIXTURE["%s/step%d/%s" % (name, k, key)]
            if rtol and key == "c_max":
                assert_allclose(got, want, rtol=rtol)
            elif key == "bound" and case["dynamics"]["kind"] == "gp":
                # the confidence bound of the chosen sample is a GP posterior standard deviation:
                # the reference's GPRCached ran here with left-to-right dot products, the oracle
                # uses BLAS (tests/gp_cases.py::reference_gp_tolerance; cond(K) < 1e6 here)
                assert_allclose(got, want, rtol=1e-10, atol=0)
            else:
                assert_equal(got, want, err_msg="%s step %d %s" % (name, k, key))


def test_scenarios_are_not_vacuous():
    """The fixture holds growing, shrinking, hand-marked, exhaustive and empty safe sets, both
    early-exit positions of the c_max index arithmetic and refined cells."""
    updates = {}
    for entry in INDEX:
        name = entry["meta"]["name"]
        updates[name] = [FIXTURE["%s/step%d/safe_set" % (name, k)] for k in range(entry["records"])
                         if "%s/step%d/safe_set" % (name, k) in FIXTURE.files]
    assert updates["nothing_safe"][0].sum() == 0
    assert updates["everything_safe_batch50"][0].all()
    marks = [int(s.sum()) for s in updates["pendulum_analytic_marks"]]
    assert marks[1] > marks[0] == marks[2]                  # hand-marked cells of later batches stay
    gp = [int(s.sum()) for s in updates["pendulum_gp"]]
    assert gp[1] > gp[0] > 100                          # grows after the new data
    refined = [FIXTURE["%s/step0/refinement" % e["meta"]["name"]] for e in INDEX
               if e["meta"].get("adaptive")]
    assert any((r > 1).any() for r in refined)
