"""BASELINE.json's other configurations at their full sizes (need an MI355X):

  C2  pendulum 256^2, 512-point GP        every cell against the oracle
  C2-table  pendulum 251^2, 128-point GP, table V and table policy    every cell off the table lines
  C3  pendulum 2048^2, 2048-point GP, LyapunovNetwork [64,64,64]
                                          sub-range records + masks, sampled cells, level set
  C4-det  cart-pole 128^4, Euler dynamics sampled cells, level set
  C5  cart-pole 64^4 x 9 actions          one Bellman optimality sweep and one policy-evaluation
                                          sweep on sampled vertices, monotone residual decay
Reference: lyapunov.py:407-606, reinforcement_learning.py:65-140, 213-279.
"""

import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import cases
import oracle
from conftest import ROOT
from test_gpu_full_size import (_grid_points, _level_set_properties, _neg_mask, _point_records,
                                _subrange_records)
from test_gpu_lyapunov import _check_masks, _engine_records, _oracle_all

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


def _workload(config, **over):
    import bench
    flags = ["--config", config]
    for key, val in over.items():
        flags += ["--" + key.replace("_", "-"), str(val)]
    return bench.build_workload(bench.parse_args(flags))[2]


def test_c2_pendulum_256_every_cell():
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _workload("C2")
    assert case["num_points"] == [256, 256] and len(case["dynamics"]["X"]) == 512
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    values, neg, rec = _engine_records(lyap)
    ref_rec, ref_neg = _oracle_all(olyap)
    assert_array_equal(values, olyap.values)
    assert_allclose(rec[:, 2:4], ref_rec[:, 2:4], rtol=1e-9, atol=1e-12)
    assert_allclose(rec[:, 4:], ref_rec[:, 4:], rtol=1e-7, atol=1e-12)
    assert_allclose(rec[:, :2], ref_rec[:, :2], rtol=1e-7, atol=1e-12)
    flips, _ = _check_masks(neg, ref_neg, rec, ref_rec)
    assert ref_neg.sum() > 1000 and (~ref_neg).sum() > 1000
    lyap.update_safe_set(); olyap.update_safe_set()
    init = np.count_nonzero(cases.initial_safe_mask(case))
    assert olyap.safe_set.sum() >= init + 1000
    if flips:
        # cells within rounding of the threshold decided differently: the oracle's sequential rule
        # on the ENGINE's decrease mask - compared modulo the flipped cells, never skipped
        grid = olyap.discretization
        olyap.negative = lambda states: neg[grid.state_to_index(states)]
        olyap.update_safe_set()
    assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max
    assert lyap.safe_count == int(olyap.safe_set.sum())


def _sampled_checks(lyap, olyap, n, neg, rtol, nsample, seed, starts=()):
    """Sub-range sweeps (records + mask bits) and scattered explicit-point records vs the oracle."""
    rng = np.random.default_rng(seed)
    starts = set(starts) | {int(s) * 64 for s in rng.integers(0, n // 64 - 32, 4)}
    both = 0
    for lo in sorted(starts):
        lo = max(0, (lo // 64) * 64)
        hi = min(n, lo + 1536)
        rec, bits = _subrange_records(lyap, lo, hi)
        ref = cases.oracle_cell_records(olyap, np.arange(lo, hi))
        assert_allclose(rec, ref, rtol=rtol, atol=1e-12)
        ref_neg = ref[:, 0] < ref[:, 1]
        margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
        assert not np.any((bits != ref_neg) & (margin > 1e-8))
        assert_array_equal(bits, neg[lo:hi])
        both += int(ref_neg.any() and (~ref_neg).any())
    idx = np.unique(rng.integers(0, n, nsample))
    ref = cases.oracle_cell_records(olyap, idx)
    assert_allclose(_point_records(lyap, idx), ref, rtol=rtol, atol=1e-12)
    ref_neg = ref[:, 0] < ref[:, 1]
    margin = np.abs(ref[:, 0] - ref[:, 1]) / np.maximum(np.abs(ref[:, 0]), 1e-300)
    assert not np.any((neg[idx] != ref_neg) & (margin > 1e-8))
    assert ref_neg.any() and (~ref_neg).any()
    return both


def test_c2_table_251_every_cell():
    """The notebooks' table-V / table-policy sweep (bench config C2-table, inverted_pendulum.ipynb
    cell 14) on its 251 x 251 grid: every cell against the oracle, away from the table grid's
    lines (where the reference's own answer depends on scipy's search history)."""
    from safe_learning_amd.benchmarks import build_lyapunov
    from test_gpu_lyapunov import _on_table_face
    case = _workload("C2-table")
    assert case["num_points"] == [251, 251] and case["V"]["kind"] == "table"
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    values, neg, rec = _engine_records(lyap)
    ref_rec, ref_neg = _oracle_all(olyap)
    otri, opol = olyap.lyapunov_function, olyap.policy.fun
    states = olyap.discretization.all_points
    ok_x = ~(_on_table_face(otri, states) | _on_table_face(opol, states))
    both = ok_x & ~_on_table_face(otri, ref_rec[:, 2:4])
    assert both.sum() > 25000
    assert_allclose(values[ok_x], olyap.values[ok_x], rtol=1e-12, atol=1e-14)
    assert_allclose(rec[ok_x][:, 2:], ref_rec[ok_x][:, 2:], rtol=1e-9, atol=1e-12)
    assert_allclose(rec[ok_x][:, 1], ref_rec[ok_x][:, 1], rtol=1e-9, atol=1e-14)
    assert_allclose(rec[both][:, 0], ref_rec[both][:, 0], rtol=1e-7, atol=1e-12)
    assert ref_neg[both].sum() > 1000 and (~ref_neg[both]).sum() > 1000
    _check_masks(neg[both], ref_neg[both], rec[both], ref_rec[both])


def test_c3_pendulum_2048_network():
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _workload("C3")
    assert case["num_points"] == [2048, 2048] and len(case["dynamics"]["X"]) == 2048
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    lyap.update_safe_set()
    neg = _neg_mask(lyap)
    assert neg.sum() > 10000 and (~neg).sum() > 10000
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    # V of the network on sampled grid points (north_star: within 1e-5; here 1e-12)
    rng = np.random.default_rng(5)
    idx = np.unique(rng.integers(0, n, 20000))
    v_ref = np.ravel(olyap.lyapunov_function(_grid_points(olyap.discretization, idx)))
    assert_allclose(lyap.values[idx], v_ref, rtol=1e-12, atol=1e-15)
    centre = (n // 2 + 1024) // 64 * 64
    both = _sampled_checks(lyap, olyap, n, neg, 1e-8, 4000, 6, starts=(0, centre, centre + 2048 * 40))
    assert both >= 1
    # level-set structure of the whole result.  The sublevel sets of a randomly initialised
    # network (Xavier-uniform weights) are not invariant under the dynamics, so the
    # safe set stays the initial set here; test_network_level_set_grows covers a network whose
    # level set grows.
    _level_set_properties(lyap, neg)


def test_c4_det_cartpole_128_euler():
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _workload("C4-det")
    lyap = build_lyapunov(case)
    n = lyap.discretization.nindex
    assert n == 128 ** 4
    lyap.update_safe_set()
    neg = _neg_mask(lyap)
    assert neg.sum() > 10 ** 6 and (~neg).sum() > 10 ** 6
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    init = np.zeros(n, dtype=bool)
    init[lyap._initial_safe_set] = True
    assert int((lyap.safe_set & ~init).sum()) >= 100
    _level_set_properties(lyap, neg)
    failing = ~(neg | init)
    i_star = int(np.flatnonzero(failing & (lyap.values == lyap.values[failing].min()))[0])
    both = _sampled_checks(lyap, olyap, n, neg, 1e-10, 30000, 7,
                           starts=(0, (i_star // 64) * 64 - 512, n // 2, n - 2048))
    assert both >= 1


def test_c5_cartpole_64_bellman_sweeps():
    import scipy.linalg
    import bench
    import exclusions
    from test_gpu_rl import ambiguous_points
    case = _workload("C5")
    assert case["num_points"] == [64] * 4 and len(case["dynamics"]["X"]) == 1024
    rl, actions = bench.build_policy_iteration(case)
    residuals = [rl.value_iteration(actions) for _ in range(3)]
    table = rl.value_function.parameters.copy()            # input of the sweep under test
    residuals.append(rl.value_iteration(actions))
    new = rl.value_function.parameters[:, 0].copy()
    greedy = rl.policy.parameters[:, 0].copy()

    ogrid = oracle.GridWorld(case["limits"], case["num_points"])
    ovf = oracle.Triangulation(ogrid, table, project=True)
    opolicy, odynamics, _, _ = cases.oracle_specs(case)
    d = case["d"]
    oreward = oracle.QuadraticFunction(-scipy.linalg.block_diag(0.1 * np.eye(d), 0.1 * np.eye(1)))
    orl = oracle.PolicyIteration.__new__(oracle.PolicyIteration)     # skip all_points (0.5 GB)
    orl.dynamics, orl.reward_function, orl.value_function = odynamics, oreward, ovf
    orl.gamma, orl.policy = 0.98, opolicy
    n = ogrid.nindex
    rng = np.random.default_rng(9)
    idx = np.unique(np.concatenate([rng.integers(0, n, 2500), [0, n - 1, n // 2]]))
    x = _grid_points(ogrid, idx)                        # state_space rows (all_points convention)
    q = np.empty((len(idx), len(actions)))
    ok = np.ones(len(idx), dtype=bool)
    faces = np.zeros(len(idx), dtype=bool)
    for a, action in enumerate(actions[:, 0]):
        u = np.full((len(x), 1), action)
        q[:, a] = orl.future_values(x, actions=u)[:, 0]
        nxt = odynamics(x, u)[0]
        ok &= ~ambiguous_points(ovf, nxt)
        faces |= exclusions.on_boundary_face(ovf, nxt)
    # successors projected onto the boundary faces of the value grid (project=True) are COMPARED
    exclusions.report("C5 64^4 max sweep", ok, "successor", faces=faces)
    assert (faces & ok).mean() > 0.05
    assert_allclose(new[idx][ok], q.max(axis=1)[ok], rtol=1e-9, atol=1e-12)
    top2 = np.sort(q, axis=1)[:, -2:]
    tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    assert not np.any((greedy[idx] != actions[np.argmax(q, axis=1), 0]) & ok & ~tie)
    assert len(np.unique(greedy)) > 2                   # the greedy policy is not trivial

    # policy-evaluation sweep (the reference's value_iteration()) with the greedy table policy
    table2 = rl.value_function.parameters.copy()
    rl.value_iteration()
    evaluated = rl.value_function.parameters[:, 0]
    ovf2 = oracle.Triangulation(ogrid, table2, project=True)
    opol = oracle.Triangulation(ogrid, greedy[:, None])
    orl.value_function = ovf2
    u = opol(x)
    ref = orl.future_values(x, actions=u)[:, 0]
    label = "C5 64^4 policy evaluation (greedy table at its own vertices)"
    # vertices where the greedy table has several admissible values: one of them (nothing left out)
    amb = exclusions.check_own_vertices(label, orl, opol, x, evaluated[idx])
    ok = ~amb & ~ambiguous_points(ovf2, odynamics(x, u)[0])
    exclusions.report(label, ok | amb, "successor")
    assert_allclose(evaluated[idx][ok], ref[ok], rtol=1e-9, atol=1e-12)

    # residual of the optimality sweeps decays monotonically (gamma-contraction)
    residuals += [rl.value_iteration(actions) for _ in range(12)]
    tail = residuals[-12:]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(tail, tail[1:]))
    assert tail[-1] < tail[0]


def test_library_records_kernel_durations():
    """``sl_timing_configure`` / ``sl_timing_collect`` (what bench.py's ``kernel_ms`` / ``finalize_ms``
    come from): one duration per recorded call and channel, at most ``slots``, none when off."""
    from safe_learning_amd.benchmarks import build_lyapunov
    lyap = build_lyapunov(cases.make_case("pendulum", num_points=[64, 64], dynamics="linear"))
    ctx = lyap._ctx
    lyap.update_safe_set()
    assert ctx.timing_collect(ctx.TIMING_LYAP_SWEEP) == []
    ctx.timing_configure(3)
    for _ in range(5):
        lyap.update_safe_set()
    sweeps, passes = ctx.timing_collect(ctx.TIMING_LYAP_SWEEP), ctx.timing_collect(ctx.TIMING_FINALIZE)
    assert len(sweeps) == 3 and len(passes) == 3
    assert all(0.0 < ms < 50.0 for ms in sweeps + passes)
    assert ctx.timing_collect(ctx.TIMING_LYAP_SWEEP) == []          # emptied
    lyap.update_safe_set()
    assert len(ctx.timing_collect(ctx.TIMING_LYAP_SWEEP)) == 1      # ... and recording again
    assert ctx.timing_collect(ctx.TIMING_BELLMAN) == []
    ctx.timing_configure(0)
    lyap.update_safe_set()
    assert ctx.timing_collect(ctx.TIMING_LYAP_SWEEP) == []
    safe = lyap.safe_set.copy()
    lyap.update_safe_set()
    assert_array_equal(lyap.safe_set, safe)


@pytest.mark.parametrize("flags", [
    dict(config="C2-table-large"),                                  # table flavour, factor in LDS (round 6)
    dict(config="C2-notebook"),                                     # two heads, sum-of-products kernels
    dict(config="C2", num_points=2048, n_gp=128),                   # fast path, factor in LDS
    dict(config="C4", num_points=48, n_gp=192),                     # four state dimensions, factor from L2
])
def test_sixteen_wavefront_workgroups_of_the_small_gp_kernel(flags, monkeypatch):
    """Large sweeps on small training sets run ``k_gp_small`` with sixteen wavefronts per workgroup
    (four per SIMD, 128 registers; ``launch_small``): per-cell records, mask words and the failing
    key must be the very bits the 8 / 12-wavefront kernels give (``SL_GP_SMALL_WAVES=12``, the
    kernels every other test of this suite checks against the oracle)."""
    import torch
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _workload(**flags)
    outs = []
    for cap in (None, "12"):
        if cap:
            monkeypatch.setenv("SL_GP_SMALL_WAVES", cap)              # read when the context is created
        lyap = build_lyapunov(case)
        lyap._upload_model()
        n, d = lyap.discretization.nindex, lyap.discretization.ndim
        dev = lyap._ctx.torch_device
        dbg = torch.zeros((n, 2 + 2 * d), dtype=torch.float64, device=dev)
        bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        record = torch.zeros_like(lyap._d_result)
        lyap._refresh_init_bits()
        lyap._ctx.lyap_sweep(0, n, lyap._d_init, lyap._values_arg(), bits, record, dbg)
        kernel = lyap._ctx.last_kernel()
        assert "k_gp_small" in kernel and ("16 wavefronts" in kernel) == (cap is None), kernel
        outs.append((dbg, bits, record))
        del lyap
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0].view(torch.int64), outs[1][0].view(torch.int64))
    assert int((outs[0][0][:, 0] < outs[0][0][:, 1]).sum()) > 1000       # a mask that is not vacuous


@pytest.mark.parametrize("flags", [
    dict(config="C2-table"),                                        # table flavour, 985 tiles
    dict(config="C2-notebook", num_points=251),                     # two heads, sum-of-products kernels
    dict(config="C2", num_points=128, n_gp=128),                    # fast path
])
def test_four_wavefront_workgroups_of_the_small_gp_kernel(flags, monkeypatch):
    """Sweeps with fewer tiles than a device of 8-wavefront workgroups has wavefronts run ``k_gp_small``
    with four wavefronts per workgroup (``launch_small``): the same bits as the 8-wavefront kernels
    (``SL_GP_SMALL_WAVES=8``)."""
    import torch
    from safe_learning_amd.benchmarks import build_lyapunov
    case = _workload(**flags)
    outs = []
    for cap in (None, "8"):
        if cap:
            monkeypatch.setenv("SL_GP_SMALL_WAVES", cap)              # read when the context is created
        lyap = build_lyapunov(case)
        lyap._upload_model()
        n, d = lyap.discretization.nindex, lyap.discretization.ndim
        dev = lyap._ctx.torch_device
        dbg = torch.zeros((n, 2 + 2 * d), dtype=torch.float64, device=dev)
        bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        record = torch.zeros_like(lyap._d_result)
        lyap._refresh_init_bits()
        lyap._ctx.lyap_sweep(0, n, lyap._d_init, lyap._values_arg(), bits, record, dbg)
        kernel = lyap._ctx.last_kernel()
        assert "k_gp_small" in kernel and ("4 wavefronts" in kernel) == (cap is None), kernel
        outs.append((dbg, bits, record))
        del lyap
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0].view(torch.int64), outs[1][0].view(torch.int64))
    assert int((outs[0][0][:, 0] < outs[0][0][:, 1]).sum()) > 100       # a mask that is not vacuous
