"""Accuracy of the device tanh (sl_tanh, csrc/sl_model.h) through a LyapunovNetwork sweep: V of a tanh network against the
oracle's (np.tanh) on grids whose pre-activations span 1e-7 .. 30.  Development probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import safe_learning_amd as sl
import oracle
from safe_learning_amd.benchmarks import network_weights
worst = 0.0
for scale in (1e-6, 1e-3, 0.05, 0.3, 1.0, 5.0, 30.0):
    lim = [[-scale, scale]] * 2
    w = network_weights(2, [64, 64, 64], seed=3)
    vals = {}
    for ns in (sl, oracle):
        grid = ns.GridWorld(lim, [257, 255])
        net = ns.LyapunovNetwork(2, [64, 64, 64], ["tanh"] * 3, eps=1e-8, weights=w)
        lyap = ns.Lyapunov(grid, net, ns.LinearSystem((np.eye(2) * 0.9, np.zeros((2, 1)))), 1.0, 1.0, 0.0,
                           ns.LinearSystem((np.zeros((1, 2)),)), initial_set=np.zeros(257 * 255, dtype=bool))
        vals[ns] = np.asarray(lyap.values).ravel()
    rel = np.abs(vals[sl] - vals[oracle]) / np.abs(vals[oracle]).clip(1e-300)
    worst = max(worst, rel.max())
    print("scale %g: max relative difference of V %.3g (V in [%.3g, %.3g])" % (scale, rel.max(), vals[oracle].min(), vals[oracle].max()))
print("worst", worst)
