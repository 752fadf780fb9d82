"""Host logic of tools/shard_balance.py (the one-GPU proxy of the multi-GPU run)."""

import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import shard_balance  # noqa: E402


def test_cost_weighted_bounds_tile_the_grid():
    """Cost-weighted cuts stay contiguous, 64-aligned and cover [0, n) exactly - for flat, skewed and
    degenerate cost profiles, also with more ranks than aligned pieces."""
    rng = np.random.default_rng(0)
    for n in (64, 150, 4096, 128 ** 3 + 17):
        for slabs in (1, 3, 8):
            cells = np.full(slabs, n // slabs)
            cells[-1] += n - cells.sum()
            for cost in (np.ones(slabs), rng.random(slabs) + 0.01, np.r_[np.zeros(slabs - 1), 1.0]):
                for world in (1, 2, 3, 8):
                    b = shard_balance.weighted_bounds(n, world, cost * cells, cells)
                    assert len(b) == world + 1 and b[0] == 0 and b[-1] == n
                    assert all(x <= y for x, y in zip(b[:-1], b[1:]))
                    assert all(x % 64 == 0 for x in b[1:-1])
    # equal cost per cell reproduces (up to alignment) the equal-cell cut
    n, world = 128 ** 4, 8
    cells = np.full(8, n // 8)
    b = shard_balance.weighted_bounds(n, world, np.ones(8) * cells, cells)
    from safe_learning_amd.distributed import shard_bounds
    assert b == shard_bounds(n, world)
    # a slab twice as expensive gets half the cells of the others (2 ranks, 2 slabs)
    b = shard_balance.weighted_bounds(6400, 2, np.array([2.0, 1.0]) * 3200, np.array([3200, 3200]))
    assert abs(b[1] - 2400) <= 64


def test_kink_runs_against_brute_force():
    from safe_learning_amd.benchmarks import headline_case
    from safe_learning_amd.distributed import shard_bounds
    case = headline_case(num_points=12, n_gp=16)
    n = 12 ** 4
    bounds = shard_bounds(n, 3)
    got, total, runs = shard_balance.kink_runs(case, bounds)
    axes = [np.linspace(-1, 1, 12)] * 4
    x = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 4)
    u = (x @ np.asarray(case["K"]).reshape(-1)).reshape(-1, 12)       # one grid row per line: a 12-cell run
    kink = ((u[:, 0] < -1) != (u[:, -1] < -1)) | ((u[:, 0] > 1) != (u[:, -1] > 1))
    starts = np.arange(0, n, 12)
    want = [int(kink[(starts >= a) & (starts < b)].sum()) for a, b in zip(bounds[:-1], bounds[1:])]
    assert got == want and total == sum(want) and runs == len(starts)
