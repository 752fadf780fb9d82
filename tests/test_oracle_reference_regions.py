"""``get_lyapunov_region`` against the REFERENCE'S OWN function, run in the build container.

``tests/golden/reference_regions.npz`` was computed by ``lyapunov.py:59-139`` executed unmodified
(``tests/golden/make_reference_regions.py``: ``tiebreaker.next()``, ``np.bool`` and the NumPy-1
reading of a list of index arrays are given back to it) on piecewise-linear landscapes with several
basins, in one to four dimensions.  Until round 3 this function was "parity unpinned" (Python-2
code).  Here, without a GPU:

* the oracle's restatement (``oracle/np_lyapunov.py``: the heap flood) reproduces every region, and
  the oracle's table evaluation reproduces the value tables the reference computed them from;
* the PARALLEL formulation the engine uses (``sl_region.hip``: minimax-distance fixpoint, stop level,
  last pop and descent), restated in NumPy (``np_shard_engine.minimax_region``), gives the same
  regions - on the fixture and on seeded random landscapes against the oracle's heap flood.
The GPU test (``tests/test_gpu_regions.py``) compares the kernels with the same fixture.
"""

import os

import numpy as np
import pytest

import oracle
from oracle import np_lyapunov

from np_shard_engine import minimax_region

FIXTURE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_regions.npz"))
NAMES = [str(n) for n in FIXTURE["_names"]]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_region_equals_the_reference_run(name):
    grid = oracle.GridWorld(FIXTURE[name + "/limits"], FIXTURE[name + "/num_points"])
    table = oracle.Triangulation(oracle.GridWorld(FIXTURE[name + "/limits"], FIXTURE[name + "/table_points"]),
                                 FIXTURE[name + "/vertex_values"][:, None], project=True)
    values = table(grid.all_points)[:, 0]
    assert np.array_equal(values, FIXTURE[name + "/values"])
    region = np_lyapunov.get_lyapunov_region(table, grid, tuple(FIXTURE[name + "/init_node"]))
    assert np.array_equal(region, FIXTURE[name + "/region"])


@pytest.mark.parametrize("name", NAMES)
def test_parallel_formulation_equals_the_reference_run(name):
    shape = tuple(int(v) for v in FIXTURE[name + "/num_points"])
    region = minimax_region(FIXTURE[name + "/values"].reshape(shape), tuple(FIXTURE[name + "/init_node"]))
    assert np.array_equal(region, FIXTURE[name + "/region"])


def test_fixture_is_not_vacuous():
    sizes = {name: int(FIXTURE[name + "/region"].sum()) for name in NAMES}
    assert sizes["2d_start_on_boundary"] == 0 and sizes["2d_start_off_minimum"] == 2
    assert sizes["2d_bowl_smooth"] > 1000 and 10 < sizes["2d_bumps_deep"] < 100      # boundary / descent
    assert {len(FIXTURE[name + "/num_points"]) for name in NAMES} == {1, 2, 3, 4}


@pytest.mark.parametrize("seed", range(12))
def test_parallel_formulation_equals_the_heap_flood_on_random_landscapes(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 4))
    shape = tuple(int(v) for v in rng.integers(6, 18, d)) if d > 1 else (int(rng.integers(20, 60)),)
    axes = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    bowl = sum((a - rng.uniform(-0.3, 0.3)) ** 2 for a in axes)
    values = bowl + rng.uniform(0.0, 0.6) * rng.random(shape)          # rough: many small basins
    assert len(np.unique(values)) == values.size
    grid = oracle.GridWorld([[-1.0, 1.0]] * d, list(shape))
    inner = values[tuple(slice(1, -1) for _ in shape)]
    starts = [tuple(int(v) + 1 for v in np.unravel_index(np.argmin(inner), inner.shape)),
              tuple(int(rng.integers(1, n - 1)) for n in shape), tuple(int(rng.integers(0, n)) for n in shape)]
    for start in starts:
        if any(s == 0 for s in start):
            continue                       # lower-boundary starts: the reference's wrap-around quirk
        want = np_lyapunov.get_lyapunov_region(lambda pts: values.reshape(-1, 1), grid, start)
        assert np.array_equal(minimax_region(values, start), want), (seed, start)
