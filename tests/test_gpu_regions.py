"""``get_lyapunov_region`` on the MI355X (``sl_region.hip``: minimax-distance fixpoint) against the
regions computed by the REFERENCE'S OWN function (``tests/golden/reference_regions.npz``,
``lyapunov.py:59-139`` run in the build container) and against the oracle's heap flood on larger
seeded landscapes."""

import os

import numpy as np
import pytest

import oracle
from oracle import np_lyapunov

pytestmark = pytest.mark.gpu

FIXTURE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_regions.npz"))
NAMES = [str(n) for n in FIXTURE["_names"]]


@pytest.mark.parametrize("name", NAMES)
def test_region_equals_the_reference_run(name):
    import safe_learning_amd as sl
    limits, num_points = FIXTURE[name + "/limits"], [int(v) for v in FIXTURE[name + "/num_points"]]
    grid = sl.GridWorld(limits, num_points)
    table = sl.Triangulation(sl.GridWorld(limits, [int(v) for v in FIXTURE[name + "/table_points"]]),
                             FIXTURE[name + "/vertex_values"][:, None], project=True)
    region = sl.get_lyapunov_region(table, grid, tuple(int(v) for v in FIXTURE[name + "/init_node"]))
    assert region.shape == tuple(num_points) and region.dtype == bool
    assert np.array_equal(region, FIXTURE[name + "/region"])


@pytest.mark.parametrize("shape,rough,seed", [((301, 257), 0.02, 1), ((301, 257), 0.3, 2), ((40, 37, 33), 0.1, 3),
                                               ((2001,), 0.05, 4), ((12, 11, 13, 10), 0.15, 5)])
def test_region_equals_the_heap_flood_on_larger_landscapes(shape, rough, seed):
    """A quadratic bowl (the engine's own QuadraticFunction values would tie on a symmetric grid) is not
    enough here: the values are uploaded directly - seeded rough landscapes with thousands of basins."""
    import torch
    import safe_learning_amd as sl
    from safe_learning_amd import lyapunov as L
    rng = np.random.default_rng(seed)
    d = len(shape)
    axes = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    values = sum((a - rng.uniform(-0.2, 0.2)) ** 2 for a in axes) + rough * rng.random(shape)
    assert len(np.unique(values)) == values.size
    grid, ogrid = sl.GridWorld([[-1.0, 1.0]] * d, list(shape)), oracle.GridWorld([[-1.0, 1.0]] * d, list(shape))
    inner = values[tuple(slice(1, -1) for _ in shape)]
    start = tuple(int(v) + 1 for v in np.unravel_index(np.argmin(inner), inner.shape))
    want = np_lyapunov.get_lyapunov_region(lambda pts: values.reshape(-1, 1), ogrid, start)
    helper = L.Lyapunov.__new__(L.Lyapunov)
    helper._bare_init(grid, sl.QuadraticFunction(np.eye(d)))           # (context + grid; values replaced)
    dev = helper._ctx.torch_device
    d_values = torch.from_numpy(values.reshape(-1).copy()).to(dev)
    work = torch.empty(values.size, dtype=torch.float64, device=dev)
    region = torch.empty(values.size, dtype=torch.uint8, device=dev)
    sweeps = helper._ctx.lyapunov_region(d_values, int(np.ravel_multi_index(start, shape)), work, region)
    got = region.cpu().numpy().astype(bool).reshape(shape)
    assert np.array_equal(got, want), (int(got.sum()), int(want.sum()), sweeps)
    assert want.sum() >= 2
