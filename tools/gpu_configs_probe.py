"""Timings of every BASELINE.json configuration on one MI355X (development probe, not a test).
Writes gpurun_out/configs.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.linalg
import torch

import safe_learning_amd as sl
from safe_learning_amd.benchmarks import build_lyapunov, build_specs, make_case, network_weights

results = []


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def lyap_config(label, case, reps=3):
    t0 = time.perf_counter()
    lyap = build_lyapunov(case)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    lyap._ctx.timing_configure(64)
    sec = timed(lyap.update_safe_set, reps)
    kern = float(np.mean(lyap._ctx.timing_collect(0)[1:]))
    lyap._ctx.timing_configure(0)
    n = lyap.discretization.nindex
    sec_values = timed(lyap.update_values, reps)
    r = {"config": label, "cells": n, "ms_update_safe_set": sec * 1e3, "ms_sweep_kernel": kern,
         "checks_per_s": n / sec, "ms_update_values": sec_values * 1e3, "setup_s": setup,
         "values_GBps": n * 8 / sec_values / 1e9, "sweep_GBps_algorithmic": n * 8.25 / (kern * 1e-3) / 1e9,
         "safe_cells": int(lyap.safe_set.sum()), "c_max": lyap.c_max}
    print(r, flush=True)
    results.append(r)
    return lyap


ONLY = os.environ.get("SL_CONFIGS", "").split(",") if os.environ.get("SL_CONFIGS") else None


def want(tag):
    return ONLY is None or tag in ONLY


_lyap_config = lyap_config


def lyap_config(label, case, reps=3):
    if want(label.split()[0]):
        return _lyap_config(label, case, reps)


lyap_config("C1 1-D 1001 cells, linear dynamics", make_case("1d"), reps=20)
lyap_config("C2 pendulum 256^2, 512-pt GP", make_case("pendulum", num_points=256, n_gp=512))
c3 = make_case("pendulum", num_points=2048, n_gp=2048)
c3["V"] = {"kind": "network", "layer_dims": [64, 64, 64], "activations": ["tanh"] * 3, "eps": 1e-8,
           "weights": network_weights(2, [64, 64, 64], seed=1)}
c3["lv"] = ("norm_grad",)
lyap_config("C3 pendulum 2048^2, 2048-pt GP, LyapunovNetwork [64,64,64]", c3, reps=1)
lyap_config("C3q pendulum 2048^2, 2048-pt GP, quadratic V", make_case("pendulum", num_points=2048, n_gp=2048), reps=1)
lyap_config("C4-lin cart-pole 128^4, linear dynamics", make_case("cartpole", num_points=128, dynamics="linear"), reps=3)
lyap_config("C4-det cart-pole 128^4, Euler cart-pole dynamics", make_case("cartpole", num_points=128, dynamics="analytic"), reps=3)

# C5: value iteration, 64^4 vertices x 9 actions, 1024-pt GP mean dynamics
if not want("C5"):
    with open(os.path.join(ROOT, "gpurun_out", "configs_partial.json"), "w") as f:
        json.dump(results, f, indent=1)
    sys.exit(0)
case = make_case("cartpole", num_points=64, n_gp=1024)
policy, dynamics, _, _ = build_specs(case)
grid = sl.GridWorld(case["limits"], case["num_points"])
vf = sl.Triangulation(grid, np.zeros((grid.nindex, 1)), project=True)
reward = sl.QuadraticFunction(-scipy.linalg.block_diag(0.1 * np.eye(4), 0.1 * np.eye(1)))
rl = sl.PolicyIteration(policy, dynamics, reward, vf, gamma=0.98)
actions = np.linspace(-1, 1, 9)[:, None]


def bellman_max():
    v_new, argmax, q, stats = rl._sweep(rl.policy, actions)
    rl.value_function._adopt_device_table(v_new.reshape(-1, 1).contiguous())
    return stats

sec = timed(bellman_max, 3)
r = {"config": "C5 cart-pole 64^4 x 9 actions, 1024-pt GP mean, Bellman max sweep", "cells": grid.nindex,
     "ms_per_sweep": sec * 1e3, "vertex_action_pairs_per_s": grid.nindex * 9 / sec}
print(r, flush=True); results.append(r)
sec = timed(rl.value_iteration, 3)
r = {"config": "C5p cart-pole 64^4, policy evaluation sweep (value_iteration), saturated linear policy",
     "cells": grid.nindex, "ms_per_sweep": sec * 1e3}
print(r, flush=True); results.append(r)
for _ in range(3):
    bellman_max()
rl.discrete_policy_optimization(actions)
best = np.sort(rl.policy._host_parameters().reshape(64 ** 3, 64), axis=1)
distinct = float(((np.diff(best, axis=1) != 0).sum(axis=1) + 1).mean())
sec = timed(rl.value_iteration, 3)
r = {"config": "C5g cart-pole 64^4, policy evaluation sweep, greedy 9-action table policy",
     "cells": grid.nindex, "ms_per_sweep": sec * 1e3, "distinct_actions_per_row": distinct}
print(r, flush=True); results.append(r)
rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
sec = timed(rl.value_iteration, 3)
r = {"config": "C5z cart-pole 64^4, policy evaluation sweep, constant table policy",
     "cells": grid.nindex, "ms_per_sweep": sec * 1e3, "distinct_actions_per_row": 1.0}
print(r, flush=True); results.append(r)
if os.environ.get("SL_C5_SHORT"):
    sys.exit(0)
# sweeps to convergence at max|dV| <= 1e-6 max|V| (bounded)
residuals = []
for it in range(60):
    stats = bellman_max()
    res = float(stats[0]); vmax = float(rl.value_function._device_table.abs().max())
    residuals.append(res / max(vmax, 1e-300))
    if res <= 1e-6 * vmax:
        break
results.append({"config": "C5 convergence", "sweeps": len(residuals), "last_relative_residual": residuals[-1]})
print(results[-1], flush=True)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w") as f:
    json.dump(results, f, indent=1)
