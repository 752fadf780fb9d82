"""The device primitives of the adaptive branch (sl_adaptive.hip) on an MI355X: the stable radix
sort and partition against NumPy, and the per-batch kernels on synthetic per-cell rows against the
reference's loop restated on arrays (``np_shard_engine.reference_adaptive_loop``) - inputs under which
whole batches are accepted through refinement, which the oracle's models hardly produce.  The
end-to-end parity of ``update_safe_set(max_refinement > 1)`` is ``test_gpu_lyapunov.test_adaptive_branch``,
the reference-run ``adaptive_*`` scenarios and the two-rank test of ``test_gpu_distributed``."""

import numpy as np
import pytest

from np_shard_engine import np_vbits, reference_adaptive_loop, synthetic_adaptive_cells

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from safe_learning_amd import _hip
    return _hip.Context()


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 70001, 1500037])
def test_sort_pairs_is_a_stable_ascending_sort(ctx, n):
    import torch
    from safe_learning_amd import _hip
    rng = np.random.default_rng(n)
    # value bits with many ties and all eight bytes in play
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    keys[rng.random(n) < 0.3] &= np.uint64(0xFF00FF0000FF00FF)
    keys[rng.random(n) < 0.2] = np.uint64(0x8000000000000000)
    dev = ctx.torch_device
    d_keys = torch.from_numpy(keys.view(np.int64).copy()).to(dev)
    d_vals = torch.arange(n, dtype=torch.int64, device=dev)
    counts = torch.zeros(_hip.SORT_COUNT_WORDS, dtype=torch.int32, device=dev)
    ctx.sort_pairs(n, d_keys, d_vals, torch.empty_like(d_keys), torch.empty_like(d_vals), counts)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(d_vals.cpu().numpy(), order)
    assert np.array_equal(d_keys.cpu().numpy().view(np.uint64), keys[order])


@pytest.mark.parametrize("n,ndigits", [(1, 1), (5000, 3), (70001, 8), (1500037, 2)])
def test_partition_by_digit_is_stable(ctx, n, ndigits):
    import torch
    from safe_learning_amd import _hip
    rng = np.random.default_rng(n)
    digits = rng.integers(0, ndigits, n).astype(np.uint8)
    dev = ctx.torch_device
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    buckets = torch.empty(256, dtype=torch.int64, device=dev)
    counts = torch.zeros(_hip.SORT_COUNT_WORDS, dtype=torch.int32, device=dev)
    ctx.partition_by_digit(n, torch.from_numpy(digits).to(dev), perm, buckets, counts)
    assert np.array_equal(perm.cpu().numpy(), np.argsort(digits, kind="stable"))
    assert np.array_equal(buckets.cpu().numpy(), np.bincount(digits, minlength=256))
    rows = torch.from_numpy(rng.integers(-5, 5, (n, 6))).to(dev)
    out = torch.empty_like(rows)
    ctx.gather_rows(n, 6, perm, rows, out)
    assert torch.equal(out, rows[perm])


@pytest.mark.parametrize("n,batch,max_ref,hard,seed", [
    (5000, 64, 3, 0.002, 1), (3000, 100, 3, 0.0, 2), (700, 50, 2, 0.01, 3), (4096, 256, 5, 0.001, 4),
    (200000, 10000, 3, 0.00002, 5),                 # the reference's batch size
])
def test_adaptive_kernels_on_synthetic_cells(n, batch, max_ref, hard, seed, monkeypatch):
    import torch
    import safe_learning_amd as sl
    from safe_learning_amd import _hip
    from safe_learning_amd.configuration import config
    from safe_learning_amd.lyapunov import _HipAdaptiveEngine, adaptive_rule
    values, decrease, thr0, tau, init = synthetic_adaptive_cells(n, seed, hard=hard)
    grid = sl.GridWorld([[-1.0, 1.0]], n)
    lyap = sl.Lyapunov(grid, sl.QuadraticFunction([[1.0]]), sl.LinearSystem((np.array([[1.0, 0.0]]),)), 0.0,
                       1.0, tau, sl.LinearSystem((np.array([[0.0]]),)), initial_set=init)
    lyap._refresh_init_bits()
    dev = lyap._ctx.torch_device

    class Synthetic(_HipAdaptiveEngine):
        def pack(self):
            ly = self.lyap
            rows = np.empty((n, 6), dtype=np.int64)
            rows[:, 0] = np_vbits(values).view(np.int64)
            rows[:, 1] = np.arange(n)
            rows[:, 2] = decrease.view(np.int64)
            rows[:, 3] = thr0.view(np.int64)
            if self.can_shrink:
                prior, ref = init, init.astype(np.int64)
            else:
                prior, ref = ly.safe_set.copy(), np.asarray(ly._refinement).astype(np.int64)
            rows[:, 4] = ref
            rows[:, 5] = init * 1 + prior * 2
            return torch.from_numpy(rows).to(dev)

    prior_safe = prior_ref = None
    for shrink in (True, False):
        p_safe = init if shrink else prior_safe
        p_ref = init.astype(np.int64) if shrink else prior_ref
        want = reference_adaptive_loop(values, decrease, thr0, tau, init, p_safe, p_ref, batch, max_ref, 1.1)
        engine = Synthetic(lyap, shrink)
        stats = {}
        c_max = adaptive_rule(engine, n, batch, max_ref, 1.1, stats)
        lyap._publish_refinement(engine.refinement)
        lyap._safe_host_valid, lyap._safe_dev_valid = False, True
        assert np.array_equal(lyap.safe_set, want[0])
        assert np.array_equal(np.asarray(lyap._refinement), want[1])
        assert c_max == want[2] or (np.isnan(c_max) and np.isnan(want[2]))
        assert stats["safe"] == int(want[0].sum())
        prior_safe, prior_ref = want[0].copy(), want[1].copy()
        marks = np.random.default_rng(seed).choice(n, 30)
        prior_safe[marks] = True                               # hand-marked cells
        lyap.safe_set[marks] = True
        lyap._safe_host_valid, lyap._safe_dev_valid = True, False
        lyap._refinement_host = np.asarray(lyap._refinement).copy()
        lyap._refinement_dev = None
