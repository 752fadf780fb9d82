"""BASELINE.json's full grid sizes, where the oracle cannot sweep every cell: parity on random
samples of cells plus size-independent properties of the result (level-set structure of the safe
set, idempotence, independence of the sharding)."""

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases

pytestmark = pytest.mark.gpu


def _neg_mask(lyap):
    n = lyap._hi - lyap._lo
    bits = lyap._d_neg.cpu().numpy().view(np.uint8)
    return np.unpackbits(bits, bitorder="little")[:n].astype(bool)


def _level_set_properties(lyap, neg):
    """can_shrink=True: safe = init | {key < key*} with key* the smallest failing (V, index)."""
    values, safe = lyap.values, lyap.safe_set
    init = np.zeros(len(values), dtype=bool)
    init[lyap._initial_safe_set] = True
    failing = ~(neg | init)
    assert failing.any()
    v_star = values[failing].min()
    i_star = np.flatnonzero(failing & (values == v_star))[0]
    below = values < v_star
    ties = np.flatnonzero(values == v_star)
    below[ties[ties < i_star]] = True
    assert_array_equal(safe, init | below)
    grown = safe & ~init
    assert neg[grown].all()                       # every cell that was added passed the check
    assert lyap.c_max == values[below].max()


def test_cartpole_128_linear_dynamics():
    """Config C4 with deterministic dynamics: bit-exact against the oracle on 40 000 random cells."""
    from safe_learning_amd.benchmarks import build_lyapunov
    case = cases.make_case("cartpole", num_points=128, dynamics="linear", tau_scale=0.004)
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    assert n == 128 ** 4
    lyap.update_safe_set()
    neg = _neg_mask(lyap)
    safe = lyap.safe_set.copy()
    _level_set_properties(lyap, neg)
    # sampled parity: records and mask bit for bit
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    rng = np.random.default_rng(0)
    idx = np.unique(np.concatenate([rng.integers(0, n, 40000), np.flatnonzero(safe)[:5000],
                                    [0, n - 1, 127, 128 ** 3 - 1]]))
    states = olyap.discretization.index_to_state(idx)
    assert_array_equal(neg[idx], olyap.negative(states))
    assert_array_equal(lyap.values[idx],
                       np.ravel(olyap.lyapunov_function(_grid_points(olyap.discretization, idx))))
    # idempotence and monotone growth
    c_max = lyap.c_max
    lyap.update_safe_set()
    assert_array_equal(lyap.safe_set, safe)
    assert lyap.c_max == c_max
    lyap.update_safe_set(can_shrink=False)
    assert (lyap.safe_set | ~safe).all()
    # the sweep over two sub-ranges writes the same mask words as the sweep over the whole grid
    import torch
    whole = lyap._d_neg.clone()
    mid = (n // 2 // 64 + 7) * 64
    lyap._d_neg.zero_()
    lyap._ctx.lyap_sweep(0, mid, lyap._d_init, lyap._d_values, lyap._d_neg, lyap._d_result)
    lo_words = lyap._d_neg[:mid // 64].clone()
    part = torch.zeros_like(lyap._d_neg)
    lyap._ctx.lyap_sweep(mid, n, lyap._d_init[mid // 64:], lyap._d_values[mid:], part, lyap._d_result)
    assert torch.equal(lo_words, whole[:mid // 64])
    assert torch.equal(part[:(n - mid) // 64], whole[mid // 64:n // 64])


def _grid_points(grid, idx):
    """all_points rows of a few flat indices without materialising the whole array."""
    ijk = np.stack(np.unravel_index(idx, grid.num_points), axis=1)
    pts = ijk * grid.unit_maxes + grid.offset
    last = ijk == (np.asarray(grid.num_points) - 1)
    return np.where(last, np.asarray(grid.limits)[:, 1], pts)


def test_cartpole_64_gp_dynamics():
    """64^4 cells with the 1024-point GP of the headline config: sampled parity within 1e-9
    (required: 1e-5) and the level-set structure of the result."""
    from safe_learning_amd.benchmarks import build_lyapunov
    from test_gpu_lyapunov import _engine_records
    case = cases.make_case("cartpole", num_points=64, n_gp=1024, tau_scale=0.0)
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    values, neg, rec = _engine_records(lyap)
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    rng = np.random.default_rng(1)
    idx = np.unique(rng.integers(0, n, 6000))
    ref = cases.oracle_cell_records(olyap, idx)
    assert_allclose(rec[idx], ref, rtol=1e-9, atol=1e-12)
    ref_neg = olyap.negative(olyap.discretization.index_to_state(idx))
    margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
    differs = neg[idx] != ref_neg
    assert not np.any(differs & (margin > 1e-9))
    lyap.update_safe_set()
    _level_set_properties(lyap, _neg_mask(lyap))


def test_cartpole_128_gp_dynamics():
    """The headline workload itself (bench.py: 128^4 cells, 1024-point GP): one full
    update_safe_set, mask parity on 5 000 random cells, level-set structure of the result."""
    from safe_learning_amd.benchmarks import build_lyapunov
    case = cases.make_case("cartpole", num_points=128, n_gp=1024)
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    lyap.update_safe_set()
    neg = _neg_mask(lyap)
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    rng = np.random.default_rng(2)
    idx = np.unique(np.concatenate([rng.integers(0, n, 5000), np.flatnonzero(lyap.safe_set)[:2000]]))
    ref = cases.oracle_cell_records(olyap, idx)
    ref_neg = olyap.negative(olyap.discretization.index_to_state(idx))
    margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
    differs = neg[idx] != ref_neg
    assert not np.any(differs & (margin > 1e-9))
    assert differs.sum() <= 2
    _level_set_properties(lyap, neg)
