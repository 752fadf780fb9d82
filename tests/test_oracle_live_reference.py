"""Reference and oracle side by side on random scenarios - only where ``/root/reference`` exists
(the build container; the GPU box has no checkout and skips).

``tests/golden/make_reference_safe_sets.py --check-live N SEED`` runs the reference's own
``Lyapunov`` (behind ``tests/golden/numpy_tf.py``) and ``oracle.Lyapunov`` on N seeded random
scenarios - grid sizes and limits, tau, batch size, linear / Euler / GP / stacked-GP dynamics, the
adaptive branch, hand-marked cells, ``can_shrink`` on and off - and requires value tables, safe sets,
``c_max`` and refinement arrays to be equal bit for bit after every call.  It runs in a process of
its own: loading the reference patches NumPy-1 aliases into the interpreter.  (A sweep of 200
scenarios, seeds 2-6, was clean when the fixtures were committed.)
"""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/safe_learning"),
                    reason="the reference checkout exists in the build container only")
@pytest.mark.parametrize("seed", [11, 12])
def test_oracle_equals_the_reference_on_random_scenarios(seed):
    script = os.path.join(ROOT, "tests", "golden", "make_reference_safe_sets.py")
    res = subprocess.run([sys.executable, script, "--check-live", "8", str(seed)], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "LIVE OK" in res.stdout, res.stdout[-3000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/safe_learning"),
                    reason="the reference checkout exists in the build container only")
@pytest.mark.parametrize("seed", [21, 22])
def test_oracle_policy_iteration_equals_the_reference_on_random_scenarios(seed):
    """The same for ``PolicyIteration`` (``make_reference_policy_iteration.py --check-live``): random
    value / policy grids (equal or not), action sets, discounts, rewards, constraint callbacks,
    Lyapunov penalties, ``bellmann_error``.  (120 scenarios, seeds 1-4, were clean when committed;
    the first sweep found the one difference there was: ``bellmann_error`` summed pairwise.)"""
    script = os.path.join(ROOT, "tests", "golden", "make_reference_policy_iteration.py")
    res = subprocess.run([sys.executable, script, "--check-live", "8", str(seed)], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "LIVE OK" in res.stdout, res.stdout[-3000:]
