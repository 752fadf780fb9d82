"""Error behaviour of the C ABI and of the host classes on a GPU box: every misuse must come back
as an error code + message (HipEngineError on the Python side), never as a wrong result."""

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sl():
    import safe_learning_amd
    return safe_learning_amd


def test_passes_require_a_model(sl):
    import torch
    from safe_learning_amd import _hip
    ctx = _hip.Context()
    dev = ctx.torch_device
    bits = torch.zeros(4, dtype=torch.int64, device=dev)
    vals = torch.zeros(64, dtype=torch.float64, device=dev)
    res = torch.zeros(8, dtype=torch.int64, device=dev)
    with pytest.raises(_hip.HipEngineError, match="sl_model_set"):
        ctx.values(0, 64, vals)
    with pytest.raises(_hip.HipEngineError, match="sl_model_set"):
        ctx.lyap_sweep(0, 64, None, vals, bits, res)
    stats = torch.zeros(2, dtype=torch.float64, device=dev)
    with pytest.raises(_hip.HipEngineError, match="sl_model_set"):
        ctx.bellman_sweep(0, 64, None, vals, None, None, stats)
    ctx.close()


def test_bad_ranges_and_limits(sl):
    import torch
    from safe_learning_amd import _hip
    from safe_learning_amd.benchmarks import build_lyapunov
    lyap = build_lyapunov(cases.make_case("pendulum", num_points=20, dynamics="linear"))
    ctx, n = lyap._ctx, lyap.discretization.nindex
    with pytest.raises(_hip.HipEngineError, match="multiple of 64"):
        ctx.lyap_sweep(3, n, lyap._d_init, lyap._d_values, lyap._d_neg, lyap._d_result)
    with pytest.raises(_hip.HipEngineError, match="bad range"):
        ctx.lyap_sweep(0, n + 1, lyap._d_init, lyap._d_values, lyap._d_neg, lyap._d_result)
    with pytest.raises(_hip.HipEngineError, match="NULL"):
        ctx.lyap_sweep(0, n, lyap._d_init, lyap._d_values, None, lyap._d_result)
    # network limits
    with pytest.raises(_hip.HipEngineError, match="layers"):
        ctx.network_set([2, 4, 4, 4, 4, 4], [1] * 5, [np.zeros((4, 2))] + [np.zeros((4, 4))] * 4)
    with pytest.raises(_hip.HipEngineError, match="width"):
        ctx.network_set([2, 65], [1], [np.zeros((65, 2))])
    # an empty range is legal and leaves "no failing cell" in the result block
    ctx.lyap_sweep(0, 0, lyap._d_init, lyap._d_values, lyap._d_neg, lyap._d_result)
    torch.cuda.synchronize()


def test_bellman_argument_checks(sl):
    import torch
    import scipy.linalg
    from safe_learning_amd import _hip
    from safe_learning_amd.benchmarks import build_specs
    case = cases.make_case("pendulum", num_points=9, dynamics="linear")
    policy, dynamics, _, _ = build_specs(case)
    grid = sl.GridWorld(case["limits"], case["num_points"])
    vf = sl.Triangulation(grid, np.zeros((grid.nindex, 1)), project=True)
    reward = sl.QuadraticFunction(-scipy.linalg.block_diag(np.eye(2), np.eye(1)))
    rl = sl.PolicyIteration(policy, dynamics, reward, vf, gamma=0.9)
    with pytest.raises(_hip.HipEngineError, match="actions"):
        rl.discrete_policy_optimization(np.linspace(-1, 1, 17)[:, None])      # more than 16 actions
    with pytest.raises(TypeError):
        sl.PolicyIteration(policy, dynamics, reward, sl.QuadraticFunction(np.eye(2)))
    with pytest.raises(NotImplementedError):
        rl.optimize_value_function()
    assert rl.value_iteration() >= 0.0


def test_host_class_argument_checks(sl):
    with pytest.raises(Exception):
        sl.GridWorld([[-1, 1]], [1])                      # fewer than two points per dimension
    grid = sl.GridWorld([[-1, 1], [-1, 1]], [5, 5])
    with pytest.raises(Exception):
        sl.Triangulation(grid, np.zeros(7))               # wrong number of vertex values

