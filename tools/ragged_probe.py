"""Bellman max sweep on a grid whose last axis is not a multiple of 64 cells: k_bellman4s
(SL_BELLMAN4=1, default) against k_bellman_mfma (SL_BELLMAN4=0).  python tools/ragged_probe.py 50"""
import os, sys, time
import numpy as np, scipy.linalg, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_learning_amd as sl
from safe_learning_amd.benchmarks import make_case, build_specs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
case = make_case("cartpole", num_points=n, n_gp=1024)
policy, dynamics, _, _ = build_specs(case)
grid = sl.GridWorld(case["limits"], case["num_points"])
vf = sl.Triangulation(grid, np.zeros((grid.nindex, 1)), project=True)
reward = sl.QuadraticFunction(-scipy.linalg.block_diag(0.1 * np.eye(4), 0.1 * np.eye(1)))
rl = sl.PolicyIteration(policy, dynamics, reward, vf, gamma=0.98)
actions = np.linspace(-1, 1, 9)[:, None]
def sweep():
    v_new, argmax, q, stats = rl._sweep(rl.policy, actions)
    rl.value_function._adopt_device_table(v_new.reshape(-1, 1).contiguous())
sweep(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): sweep()
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 3 * 1e3
print({"grid": "%d^4" % n, "cells": grid.nindex, "SL_BELLMAN4": os.environ.get("SL_BELLMAN4", "1"),
       "ms_per_sweep": ms, "pairs_per_s": grid.nindex * 9 / ms * 1e3, "kernel": rl._ctx.last_kernel()})
# policy evaluation with the greedy table policy of the sweeps above (k_bellman4_policy against
# k_bellman_policy_mfma: SL_BELLMAN4_POLICY=0)
rl.discrete_policy_optimization(actions)
def evaluate():
    rl.value_iteration()
evaluate(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): evaluate()
torch.cuda.synchronize()
print({"grid": "%d^4" % n, "policy evaluation ms": (time.perf_counter() - t) / 3 * 1e3,
       "SL_BELLMAN4_POLICY": os.environ.get("SL_BELLMAN4_POLICY", "1"), "kernel": rl._ctx.last_kernel()})
