"""The oracle's ``PolicyIteration`` against tables computed by the reference's own
``reinforcement_learning.py``.

``tests/golden/reference_policy_iteration.npz`` was produced in the build container by
``tests/golden/make_reference_policy_iteration.py``: the reference's ``PolicyIteration``
(``reinforcement_learning.py:46-140, 213-279``; the Lyapunov penalty through the reference's
``lyapunov.py:265-376``) executed unmodified on the reference's own ``Triangulation`` tables,
reward and dynamics objects, with ``tests/golden/numpy_tf.py`` answering the TensorFlow ops (a GP
model is the oracle's callable: gpflow is absent).  Every scenario
(parameters stored in the fixture) is replayed on ``oracle.PolicyIteration`` through the same
step driver; the value table after every ``value_iteration``, the policy table after every
``discrete_policy_optimization`` and every ``future_values`` / ``bellmann_error`` result must be
equal bit for bit.
"""

import importlib.util
import json
import os
import sys

import numpy as np
import pytest
from numpy.testing import assert_equal

import oracle

from conftest import GOLDEN_DIR


def _generator():
    sys.path.insert(0, GOLDEN_DIR)
    try:
        spec = importlib.util.spec_from_file_location(
            "make_reference_policy_iteration",
            os.path.join(GOLDEN_DIR, "make_reference_policy_iteration.py"))
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
    finally:
        sys.path.remove(GOLDEN_DIR)
    return module


GENERATOR = _generator()
FIXTURE = np.load(os.path.join(GOLDEN_DIR, "reference_policy_iteration.npz"))
INDEX = json.loads(str(FIXTURE["_index"]))
NAMES = [entry["scenario"]["name"] for entry in INDEX]


def _scenario(entry):
    scenario = GENERATOR.from_jsonable(entry["scenario"], FIXTURE)
    scenario["steps"] = [tuple(step) for step in scenario["steps"]]
    return scenario


def test_fixture_covers_the_scenarios_of_the_generator():
    assert NAMES == [s["name"] for s in GENERATOR.scenarios()]
    assert len(NAMES) >= 7


@pytest.mark.parametrize("entry", INDEX, ids=NAMES)
def test_tables_equal_the_reference_run(entry):
    scenario = _scenario(entry)
    name, case = scenario["name"], scenario["case"]
    policy, dynamics, reward, value, (lyap_value, lv) = GENERATOR.build_oracle_leaves(scenario)
    rl = oracle.PolicyIteration(policy, dynamics, reward, value, gamma=scenario["gamma"])
    lyap = None
    if scenario["lyapunov"]:
        grid = oracle.GridWorld(scenario["limits"], scenario["value_points"])
        lyap = oracle.Lyapunov(grid, lyap_value, dynamics, case["lf"], lv, case["tau"], policy)
    records = GENERATOR.replay(scenario, rl, lambda: value.parameters, lambda: policy.parameters,
                               value.discretization.all_points, lyap,
                               run=lambda op: None, evaluate_fv=lambda result: result)
    assert len(records) == entry["records"]
    for k, got in enumerate(records):
        want = FIXTURE["%s/record%d" % (name, k)]
        if case["dynamics"]["kind"] == "gp":
            assert GENERATOR.records_match(got, want, case), "%s record %d" % (name, k)
        else:
            assert_equal(got, want, err_msg="%s record %d" % (name, k))


def test_scenarios_are_not_vacuous():
    """Constraints rule actions out, the tie scenario has a tie, the penalty changes the values."""
    by_name = {entry["scenario"]["name"]: _scenario(entry) for entry in INDEX}
    sc = by_name["pendulum_gp_constraint"]
    policy, dynamics, reward, value, _ = GENERATOR.build_oracle_leaves(sc)
    rl = oracle.PolicyIteration(policy, dynamics, reward, value, gamma=sc["gamma"])
    x = value.discretization.all_points
    for kind in ("outwards", "all_but_one"):
        q, _ = rl.discrete_policy_optimization(np.linspace(-1, 1, 7)[:, None],
                                               constraint=GENERATOR.constraint_function(kind, x))
        assert np.isinf(q).any() and not np.isinf(q).all(axis=1).any()
    sc = by_name["pendulum_linear_ties"]
    policy, dynamics, reward, value, _ = GENERATOR.build_oracle_leaves(sc)
    rl = oracle.PolicyIteration(policy, dynamics, reward, value, gamma=sc["gamma"])
    q, best = rl.discrete_policy_optimization(sc["steps"][0][1])
    assert (q[:, 0] == q[:, 2]).all() and (best == 0).any() and not (best == 2).any()
    sc = by_name["pendulum_gp_lyapunov"]
    index = NAMES.index("pendulum_gp_lyapunov")
    penalised = FIXTURE["pendulum_gp_lyapunov/record%d" % (INDEX[index]["records"] - 2)]
    policy, dynamics, reward, value, _ = GENERATOR.build_oracle_leaves(sc)
    rl = oracle.PolicyIteration(policy, dynamics, reward, value, gamma=sc["gamma"])
    rl.value_iteration()
    rl.value_iteration()
    plain = rl.future_values(value.discretization.all_points)
    assert np.max(np.abs(plain - penalised)) > 1e-3


@pytest.mark.parametrize("entry", INDEX, ids=NAMES)
def test_reference_values_at_ambiguous_vertices_are_admissible(entry):
    """Where a piecewise-constant policy table is read at its own vertices, the ``%`` of
    ``functions.py:1116-1124`` wraps the query into a corner shared by several unit-cell simplices
    and SciPy's answer depends on its search history.  ``tests/exclusions.py`` enumerates the
    admissible answers (one per containing simplex); the GPU tests demand that the engine's value is
    one of them.  Here the enumeration itself is checked against the reference run: at every
    ambiguous vertex of every recorded sweep the REFERENCE's value is in the set."""
    import exclusions
    scenario = _scenario(entry)
    name, case = scenario["name"], scenario["case"]
    policy, dynamics, reward, value, (lyap_value, lv) = GENERATOR.build_oracle_leaves(scenario)
    rl = oracle.PolicyIteration(policy, dynamics, reward, value, gamma=scenario["gamma"])
    x = value.discretization.all_points
    record, checked = 0, 0
    for step in scenario["steps"]:
        kind = step[0]
        if kind == "vi":
            for _ in range(step[1]):
                want = FIXTURE["%s/record%d" % (name, record)]
                amb = exclusions.check_own_vertices("%s record %d" % (name, record), rl, policy, x, want,
                                                    also=want, rtol=1e-9, atol=1e-12)
                checked += int(amb.sum())
                value.parameters = want.copy()
                record += 1
        elif kind == "dpo":
            policy.parameters = FIXTURE["%s/record%d" % (name, record)].copy()
            record += 1
        else:
            record += 1
    assert record == entry["records"]
    if name in ("pendulum_analytic", "pendulum_gp_constraint", "random_1_pendulum_gp"):
        assert checked > 20                      # (a third of these tables' vertices are ambiguous)
