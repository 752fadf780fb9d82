"""BASELINE.json's full grid sizes, where the oracle cannot sweep every cell: parity on random
samples of cells plus size-independent properties of the result (level-set structure of the safe
set, idempotence, independence of the sharding)."""

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases
from test_gpu_lyapunov import log_mask_flips

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


def _subrange_records(lyap, lo, hi):
    """Per-cell records and mask bits of the grid sweep itself over [lo, hi) (lo 64-aligned)."""
    import torch
    d = lyap.discretization.ndim
    dev = lyap._ctx.torch_device
    dbg = torch.zeros((hi - lo, 2 + 2 * d), dtype=torch.float64, device=dev)
    bits = torch.zeros((hi - lo + 63) // 64, dtype=torch.int64, device=dev)
    result = torch.zeros_like(lyap._d_result)
    lyap._ctx.lyap_sweep(lo, hi, lyap._d_init[lo // 64:], lyap._d_values[lo:], bits, result, dbg)
    neg = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:hi - lo]
    return dbg.cpu().numpy(), neg.astype(bool)


def _point_records(lyap, idx):
    """The same records through the explicit-point entry (sl_eval_points, SL_EVAL_DECREASE)."""
    import torch
    from safe_learning_amd import _hip
    d = lyap.discretization.ndim
    dev = lyap._ctx.torch_device
    pts = torch.from_numpy(lyap.discretization.index_to_state(idx)).to(dev)
    out = torch.zeros((len(idx), 2 + 2 * d), dtype=torch.float64, device=dev)
    lyap._ctx.eval_points(_hip.EVAL_DECREASE, len(idx), pts, out)
    return out.cpu().numpy()


def _gp_full_size_checks(case, nsample, seed):
    """Shared body of the 64^4 / 128^4 GP tests: one full update_safe_set, then
      * the workload is non-degenerate (both mask classes, the level set grows by >= 100 cells),
      * records AND mask bits of the sweep on sub-ranges around key*, the origin, the corners and
        random places against the oracle (1e-9 relative; required 1e-5),
      * records of scattered cells through the explicit-point entry,
      * the level-set structure of the whole result."""
    from safe_learning_amd.benchmarks import build_lyapunov
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    d = case["d"]
    lyap.update_safe_set()
    neg = _neg_mask(lyap)
    safe = lyap.safe_set
    init = np.zeros(n, dtype=bool)
    init[lyap._initial_safe_set] = True
    assert neg.any() and (~neg).any(), "degenerate workload: one class only"
    grown = int((safe & ~init).sum())
    assert grown >= 100, "safe set grew by %d cells only" % grown
    _level_set_properties(lyap, neg)

    olyap = cases.oracle_lyapunov(case, compute_values=False)
    failing = ~(neg | init)
    values = lyap.values
    v_star = values[failing].min()
    i_star = int(np.flatnonzero(failing & (values == v_star))[0])
    rng = np.random.default_rng(seed)
    starts = {(i_star // 64) * 64, (n // 2 // 64) * 64, 0, ((n - 1) // 64) * 64 - 1024}
    starts |= {int(s) * 64 for s in rng.integers(0, n // 64 - 32, 6)}
    first_passing = int(np.flatnonzero(neg)[0])
    starts.add((first_passing // 64) * 64)
    both = 0
    for lo in sorted(starts):
        lo = max(0, lo - 512)
        hi = min(n, lo + 2048)
        rec, bits = _subrange_records(lyap, lo, hi)
        idx = np.arange(lo, hi)
        ref = cases.oracle_cell_records(olyap, idx)
        assert_allclose(rec, ref, rtol=1e-9, atol=1e-12)
        ref_neg = ref[:, 0] < ref[:, 1]
        margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
        differs = bits != ref_neg
        assert not np.any(differs & (margin > 1e-9)), "mask differs away from the threshold"
        log_mask_flips(int(differs.sum()), len(differs), float(margin.min()))
        assert_array_equal(bits, neg[lo:hi])          # sub-range sweep == whole-grid sweep
        both += int(ref_neg.any() and (~ref_neg).any())
    assert both >= 2, "the compared sub-ranges must contain passing and failing cells"

    # scattered cells: half of them inside / at the rim of the level set
    near = np.flatnonzero(safe)
    idx = np.unique(np.concatenate([rng.integers(0, n, nsample),
                                    rng.choice(near, min(len(near), nsample), replace=False),
                                    [i_star]]))
    ref = cases.oracle_cell_records(olyap, idx)
    assert_allclose(_point_records(lyap, idx), ref, rtol=1e-9, atol=1e-12)
    ref_neg = ref[:, 0] < ref[:, 1]
    margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
    differs = neg[idx] != ref_neg
    assert not np.any(differs & (margin > 1e-9))
    assert differs.sum() <= 2
    log_mask_flips(int(differs.sum()), len(differs), float(margin.min()))
    assert ref_neg.any() and (~ref_neg).any()
    assert not ref_neg[np.searchsorted(idx, i_star)] and not init[i_star]   # key* really fails
    return lyap, grown


def test_cartpole_64_gp_dynamics():
    """64^4 cells with the 1024-point GP of the headline config."""
    from safe_learning_amd.benchmarks import headline_case
    _gp_full_size_checks(headline_case(num_points=64), 3000, 1)


def test_cartpole_128_gp_dynamics():
    """The headline workload itself (bench.py: 128^4 cells, 1024-point GP)."""
    from safe_learning_amd.benchmarks import headline_case
    lyap, grown = _gp_full_size_checks(headline_case(), 2500, 2)
    assert lyap.discretization.nindex == 128 ** 4
    # idempotence; and a second update with can_shrink=False cannot lose cells
    safe, c_max = lyap.safe_set.copy(), lyap.c_max
    lyap.update_safe_set(can_shrink=False)
    assert (lyap.safe_set | ~safe).all() and lyap.c_max >= c_max


def test_cartpole_gp_function_stack_full_head_count():
    """FunctionStack of four single-output 1024-point GPs (the notebooks' style, 4x the work per
    cell) on 48^4 cells: non-degenerate, records and masks on sub-ranges vs the oracle."""
    from safe_learning_amd.benchmarks import headline_case
    case = headline_case(num_points=48, stack=True)
    _gp_full_size_checks(case, 1500, 3)


def test_headline_kernel_seeds_bit_identical_at_size(monkeypatch):
    """``k_gp_sweep4`` on the headline model (1024 training points, four panels, tiles with
    saturation kinks) at 48^4 cells: with the sequence seeds (``SL_GP4_SEEDS`` unset) and with every
    generation from the exponentials (``SL_GP4_SEEDS=0``) the mask, the failing key and the
    counters are bit for bit the same; so they are with the tiles drawn in another order (a second
    launch - the counter hands the tiles out in the order the workgroups ask)."""
    import torch
    from safe_learning_amd.benchmarks import build_lyapunov, headline_case
    case = headline_case(num_points=48)
    out = []
    for seeds in ("1", "0", "1"):
        monkeypatch.setenv("SL_GP4_SEEDS", seeds)
        lyap = build_lyapunov(case)
        lyap.update_safe_set()
        assert lyap._ctx.last_kernel().startswith("k_gp_sweep4")
        out.append((lyap._d_neg.cpu().numpy().copy(), lyap.safe_set.copy(), lyap.c_max,
                    lyap._d_result.cpu().numpy().copy()))
    for other in out[1:]:
        assert_array_equal(out[0][0], other[0])
        assert_array_equal(out[0][1], other[1])
        assert out[0][2] == other[2]
        assert_array_equal(out[0][3], other[3])
    neg = np.unpackbits(out[0][0].view(np.uint8), bitorder="little")[:48 ** 4]
    assert 0.05 < neg.mean() < 0.95 and out[0][1].sum() > 100
