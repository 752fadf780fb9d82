"""The sweep kernels' GP posterior against the REFERENCE'S OWN GP code (needs an MI355X).

``tests/golden/reference_gp_posterior.npz`` holds posterior means and confidence bounds computed by
the reference's ``GPRCached`` / ``GaussianProcess`` / ``FunctionStack`` (``functions.py:254-307,
357-546``; ``tests/golden/make_reference_gp.py``) at grid cells and at explicit points.  Here the
per-cell records ``[decrease, threshold, mean, error]`` of the GRID SWEEP itself - ``k_gp_small``
(<= 256 training points), ``k_gp_sweep4`` (more), ``k_gp_sweep`` (stacks of large heads) - are
compared with the FIXTURE (not with the oracle) at the fixture's cells, and the explicit-point
entry (``sl_eval_points``) at the training inputs and far-away points.  Tolerance:
``reference_gp_tolerance(cond(K))`` = 8 eps cond(K) (<= 1e-6 in the fixture; north star 1e-5).
"""

import os

import numpy as np
import pytest

from gp_cases import (reference_gp_build_case, reference_gp_case_list, reference_gp_model,
                      reference_gp_tolerance)

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_gp_posterior.npz")
SPECS = reference_gp_case_list()
EXPECTED_KERNEL = {3: "k_gp_small", 130: "k_gp_small", 136: "k_gp_small", 512: "k_gp_sweep4",
                   1024: "k_gp_sweep4"}


@pytest.fixture(scope="module")
def fixture():
    return np.load(FIXTURE)


def sweep_records(lyap, idx):
    """Records of the sweep kernel at the flat indices ``idx`` (one launch per 64-cell block that
    holds a queried cell; blocks are merged into runs)."""
    import torch
    d = lyap.discretization.ndim
    n = lyap.discretization.nindex
    dev = lyap._ctx.torch_device
    blocks = np.unique(idx // 64)
    out = np.empty((len(idx), 2 + 2 * d))
    runs, start = [], 0
    for k in range(1, len(blocks) + 1):
        if k == len(blocks) or blocks[k] != blocks[k - 1] + 1:
            runs.append((int(blocks[start]) * 64, min(n, (int(blocks[k - 1]) + 1) * 64)))
            start = k
    kernels = set()
    for lo, hi in runs:
        dbg = torch.zeros((hi - lo, 2 + 2 * d), dtype=torch.float64, device=dev)
        bits = torch.zeros((hi - lo + 63) // 64, dtype=torch.int64, device=dev)
        result = torch.zeros_like(lyap._d_result)
        lyap._ctx.lyap_sweep(lo, hi, lyap._d_init[lo // 64:], lyap._d_values[lo:], bits, result, dbg)
        kernels.add(lyap._ctx.last_kernel())
        sel = (idx >= lo) & (idx < hi)
        out[sel] = dbg.cpu().numpy()[idx[sel] - lo]
    return out, kernels


@pytest.mark.parametrize("spec", SPECS, ids=[s["name"] for s in SPECS])
def test_sweep_kernels_reproduce_the_reference_posterior(spec, fixture):
    import safe_learning_amd as sl
    import safe_learning_amd.functions as F
    from safe_learning_amd import _evaluate
    from safe_learning_amd.benchmarks import build_specs, initial_safe_mask
    name = spec["name"]
    case = reference_gp_build_case(spec)
    d = case["d"]
    dynamics = reference_gp_model(F, spec, case, fixture)
    policy, _, value, lv = build_specs(case)
    mask = np.array(initial_safe_mask(case), copy=True)   # an array of its own (no writable base) ...
    mask.flags.writeable = False          # ... read-only: identified by object identity, not hashed (256 MB at C4)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=mask)
    lyap._upload_model()
    lyap._refresh_init_bits()
    tol = reference_gp_tolerance(float(fixture[name + "/cond"]))
    idx = fixture[name + "/cell_index"]
    rec, kernels = sweep_records(lyap, idx)
    n_train = len(fixture[name + "/X"]) + spec.get("add_points", 0)
    if not case["stack"]:
        assert all(k.startswith(EXPECTED_KERNEL[n_train]) for k in kernels), kernels
    want_mean, want_bound = fixture[name + "/cell_mean"], fixture[name + "/cell_bound"]
    scale = np.abs(want_mean).max(axis=0)
    assert np.all(np.abs(rec[:, 2:2 + d] - want_mean) <= tol * scale), (name, kernels, "mean")
    assert np.all(np.abs(rec[:, 2 + d:] - want_bound) <= tol * want_bound), (name, kernels, "bound")
    # explicit points: the training inputs (variance collapses to the noise level) and points far
    # outside the data (variance returns to the prior)
    q = fixture[name + "/extra_inputs"]
    mean, bound = _evaluate.dynamics(dynamics, q[:, :d], q[:, d:])
    want_mean, want_bound = fixture[name + "/extra_mean"], fixture[name + "/extra_bound"]
    scale = np.abs(want_mean).max(axis=0)
    assert np.all(np.abs(mean - want_mean) <= tol * scale), (name, "points mean")
    assert np.all(np.abs(bound - want_bound) <= tol * want_bound), (name, "points bound")
