"""A NeuralNetwork as the policy (``functions.py:1663-1729``; ``examples/inverted_pendulum.ipynb:215``):
``csrc/sl_policy_net.hip`` evaluates it once per cell of a call into an action table and every kernel
runs on that table.  Against the oracle's dense-layer chain (needs an MI355X).  No test of the
reference holds a number for this class: its arithmetic is restated, parity unpinned."""

import numpy as np
import pytest
import scipy.linalg
from numpy.testing import assert_allclose, assert_array_equal

import cases
import oracle
from test_gpu_lyapunov import _check_masks, _engine_records, _oracle_all
from test_gpu_rl import ambiguous_points

pytestmark = pytest.mark.gpu


def _network_pair(sl, d, layers, nonlinearities, scale, use_bias, seed):
    net = sl.NeuralNetwork(layers, nonlinearities, output_scale=scale, use_bias=use_bias, input_dim=d, seed=seed)
    # (weights of a size that bends the policy inside the grid and saturates it near the rim)
    net.parameters = [3.0 * p if p.ndim == 2 else p + 0.1 * np.arange(p.size) for p in net.parameters]
    onet = oracle.NeuralNetwork(layers, nonlinearities, scale, use_bias, parameters=net.parameters)
    return net, onet


@pytest.mark.parametrize("d,layers,acts,scale,bias", [
    (2, [32, 32, 1], ["relu", "relu", "tanh"], 0.7, True),      # inverted_pendulum.ipynb:215
    (4, [64, 64, 1], ["tanh", "sigmoid", None], 1.3, False),    # the RL notebooks: no biases
    (1, [5, 1], ["tanh", None], 1.0, True),
    (3, [8, 2], ["relu", "tanh"], 0.5, True),                   # two action dimensions
])
def test_network_policy_at_points(d, layers, acts, scale, bias):
    import safe_learning_amd as sl
    from safe_learning_amd import _evaluate
    net, onet = _network_pair(sl, d, layers, acts, scale, bias, seed=3)
    pts = np.random.default_rng(0).uniform(-1, 1, (1000, d))
    assert_allclose(_evaluate.policy(net, pts), onet(pts), rtol=1e-12, atol=1e-14)
    sat = sl.Saturation(net, -0.2, 0.3)
    assert_allclose(_evaluate.policy(sat, pts), np.clip(onet(pts), -0.2, 0.3), rtol=1e-12, atol=1e-14)
    # new parameters are noticed; an in-place edit after touch()
    net.parameters = [0.5 * p for p in net.parameters]
    onet.parameters = [0.5 * p for p in onet.parameters]
    assert_allclose(_evaluate.policy(net, pts), onet(pts), rtol=1e-12, atol=1e-14)
    net.parameters[0][...] *= 2.0
    net.touch()
    onet.parameters[0] = 2.0 * onet.parameters[0]
    assert_allclose(_evaluate.policy(net, pts), onet(pts), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name,kw", [
    ("pendulum", dict(num_points=40, dynamics="analytic", tau_scale=0.02)),       # k_det_sweep
    ("pendulum", dict(num_points=[33, 64], n_gp=90, tau_scale=0.01)),             # k_gp_small
    ("cartpole", dict(num_points=[6, 6, 5, 64], n_gp=300, tau_scale=0.0)),        # k_gp_sweep4
    ("cartpole", dict(num_points=8, dynamics="linear", tau_scale=0.004)),         # not k_det_rows: a table policy
])
def test_update_safe_set_with_a_network_policy(name, kw):
    """The Lyapunov sweep with a saturated network policy: per-cell records against the oracle's, the
    masks by the margin rule (a policy value that differs in its last bits moves the decrease), the
    safe set and c_max modulo the flipped cells."""
    import safe_learning_amd as sl
    from gp_cases import INFORMED, TIGHT
    from safe_learning_amd.benchmarks import build_lyapunov
    if "n_gp" in kw:
        kw = dict(kw, **(TIGHT if name == "cartpole" else INFORMED))
    case = cases.make_case(name, **kw)
    d = case["d"]
    net, onet = _network_pair(sl, d, [16, 16, 1], ["relu", "tanh", "tanh"], 1.2, True, seed=5)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    lyap.policy = sl.Saturation(net, -1.0, 1.0)
    olyap.policy = oracle.Saturation(onet, -1.0, 1.0)
    values, neg, rec = _engine_records(lyap)
    assert "k_det_rows" not in lyap._ctx.last_kernel()
    ref_rec, ref_neg = _oracle_all(olyap)
    assert_allclose(rec[:, 2:], ref_rec[:, 2:], rtol=1e-7, atol=1e-12)
    assert_allclose(rec[:, :2], ref_rec[:, :2], rtol=1e-7, atol=1e-11)
    flips, _ = _check_masks(neg, ref_neg, rec, ref_rec)
    assert ref_neg.any() and (~ref_neg).any()
    u = onet(olyap.discretization.all_points)
    assert (np.abs(u) > 1).any() and (np.abs(u) < 1).any()          # the saturation bites somewhere
    lyap.update_safe_set()
    if flips:
        grid = olyap.discretization
        olyap.negative = lambda states: neg[grid.state_to_index(states)]
    olyap.update_safe_set()
    assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max


def test_policy_evaluation_sweep_with_a_network_policy():
    """PolicyIteration.value_iteration() / bellmann_error() / future_values() with a network policy
    (reinforcement_learning.py:65-140): the policy-evaluation sweep on its action table."""
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs
    for name, kw, nv in (("pendulum", dict(n_gp=70), [12, 64]), ("pendulum", dict(dynamics="analytic"), 21),
                         ("cartpole", dict(n_gp=90), 5)):
        case = cases.make_case(name, num_points=nv, **kw)
        d = case["d"]
        net, onet = _network_pair(sl, d, [16, 1], ["tanh", "tanh"], 0.8, True, seed=7)
        _, dynamics, _, _ = build_specs(case)
        _, odynamics, _, _ = cases.oracle_specs(case)
        qmat = -scipy.linalg.block_diag(np.eye(d), 0.1 * np.eye(1))
        vgrid, ovgrid = sl.GridWorld(case["limits"], nv), oracle.GridWorld(case["limits"], nv)
        v0 = -np.random.default_rng(4).random((vgrid.nindex, 1))
        vf, ovf = sl.Triangulation(vgrid, v0, project=True), oracle.Triangulation(ovgrid, v0, project=True)
        rl = sl.PolicyIteration(net, dynamics, sl.QuadraticFunction(qmat), vf, gamma=0.95)
        orl = oracle.PolicyIteration(onet, odynamics, oracle.QuadraticFunction(qmat), ovf, gamma=0.95)
        x = orl.state_space
        nxt = orl.dynamics(x, onet(x))
        ok = ~ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
        assert ok.mean() > 0.9
        assert_allclose(rl.future_values()[ok], orl.future_values(x)[ok], rtol=1e-9, atol=1e-12)
        for _ in range(2):
            vf.parameters = ovf.parameters.copy()
            rl.value_iteration()
            orl.value_iteration()
            assert_allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12)
        # a max sweep ignores the policy; afterwards the greedy table replaces the network
        actions = np.linspace(-1, 1, 5)[:, None]
        rl.value_iteration(actions)
        assert isinstance(rl.policy, sl.Triangulation)


def test_network_policy_errors():
    import safe_learning_amd as sl
    from safe_learning_amd._hip import HipEngineError
    from safe_learning_amd.benchmarks import build_lyapunov
    case = cases.make_case("pendulum", num_points=20, dynamics="linear")
    lyap = build_lyapunov(case)
    lyap.policy = sl.NeuralNetwork([4, 1], ["tanh", None], input_dim=3)      # the grid has 2 dimensions
    with pytest.raises(ValueError):
        lyap.update_safe_set()
    with pytest.raises(TypeError):
        sl.NeuralNetwork([4, 1], ["elu", None])
    with pytest.raises(ValueError):
        sl.NeuralNetwork([4, 1], ["tanh", None], input_dim=2, parameters=[np.zeros((2, 4))])
    wide = sl.NeuralNetwork([65, 1], ["tanh", None], input_dim=2)
    lyap.policy = wide
    with pytest.raises(HipEngineError):
        lyap.update_safe_set()
