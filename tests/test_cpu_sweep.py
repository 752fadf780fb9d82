"""oracle/cpu_sweep.cpp (the C++ / OpenMP CPU baseline of bench.py) is measurement
infrastructure: here the NumPy oracle - the checker - checks it.  No GPU needed."""

import numpy as np
import pytest

import cases
from gp_cases import INFORMED, TIGHT
from oracle import cpu_sweep


@pytest.mark.parametrize("name,kw", [
    ("pendulum", dict(num_points=40, n_gp=70, tau_scale=0.01, **INFORMED)),
    ("pendulum", dict(num_points=[33, 50], n_gp=130, tau_scale=0.01, **INFORMED)),   # n % 4 != 0
    ("cartpole", dict(num_points=[6, 7, 6, 9], n_gp=203, tau_scale=0.0, **TIGHT)),
])
def test_cpp_batch_loop_against_the_numpy_oracle(name, kw):
    case = cases.make_case(name, **kw)
    olyap = cases.oracle_lyapunov(case)
    sweep = cpu_sweep.CpuSweep(case, olyap.dynamics.gaussian_process)
    n = olyap.discretization.nindex
    idx = np.arange(n)
    ref = cases.oracle_cell_records(olyap, idx)
    for threads in (1, 3):
        neg, rec, seconds, used = sweep.check(idx, threads=threads, records=True)
        assert used == threads and seconds > 0
        scale = np.abs(ref[:, 0]).max()
        assert np.abs(rec[:, 0] - ref[:, 0]).max() <= 1e-11 * scale          # decrease
        np.testing.assert_allclose(rec[:, 1], ref[:, 1], rtol=1e-13, atol=1e-300)   # threshold
        d = case["d"]
        assert np.abs(rec[:, 2:2 + d] - ref[:, 2:2 + d]).max() <= 1e-11 * np.abs(ref[:, 2:2 + d]).max()
        np.testing.assert_allclose(rec[:, 2 + d:], ref[:, 2 + d:], rtol=1e-8)       # beta sqrt(var)
        ref_neg = ref[:, 0] < ref[:, 1]
        margin = np.abs(ref[:, 0] - ref[:, 1]) <= 1e-10 * scale
        assert not np.any((neg != ref_neg) & ~margin)
        assert 0 < neg.sum() < n                                             # cells pass AND fail
    # an index list that is no multiple of the 32-cell tile, in another order
    part = np.random.default_rng(0).permutation(n)[:77]
    neg_p, rec_p, _, _ = sweep.check(part, records=True)
    np.testing.assert_array_equal(rec_p, rec[part])
    np.testing.assert_array_equal(neg_p, neg[part])


def test_configurations_it_does_not_restate_are_refused():
    case = cases.make_case("pendulum", num_points=10, n_gp=20, stack=True)
    olyap = cases.oracle_lyapunov(case)
    with pytest.raises(ValueError):
        cpu_sweep.CpuSweep(case, olyap.dynamics.functions[0].gaussian_process)
    with pytest.raises(ValueError):
        cpu_sweep.CpuSweep(cases.make_case("pendulum", num_points=10, dynamics="linear"), None)
