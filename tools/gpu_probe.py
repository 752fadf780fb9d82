"""GPU-side probe used during development (not a test): FP64 rate probes and a timing sweep of
the GP kernel configurations.  Writes gpurun_out/probe.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from safe_learning_amd import _hip
from safe_learning_amd.benchmarks import build_lyapunov, make_case

out = {}
ctx = _hip.Context()
for per_cu in (("2",) if os.environ.get("SL_PROBE_SWEEPS") == "short" else ("1", "2", "4")):
    os.environ["SL_PROBE_BLOCKS_PER_CU"] = per_cu
    for which, name in ((0, "mfma_f64"), (1, "valu_fma_f64"), (2, "both")):
        out["rate_%s_x%s" % (name, per_cu)] = ctx.debug_fp64_rate(which, 20000)
print(out, flush=True)

def time_sweep(family, num_points, n_gp, cfg, reps=3):
    os.environ["SL_GP_CFG"] = str(cfg)
    case = make_case(family, num_points=num_points, n_gp=n_gp)
    lyap = build_lyapunov(case)
    lyap.update_safe_set()
    torch.cuda.synchronize()
    lyap._ctx.timing_configure(64)
    t0 = time.perf_counter()
    for _ in range(reps):
        lyap.update_safe_set()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms = float(np.mean(lyap._ctx.timing_collect(0)))
    n = lyap.discretization.nindex
    p, d = case["d"] + 1, case["d"]
    flops = (n_gp * (4 * p + 2) + 2 * n_gp * d + n_gp * n_gp + 2 * n_gp) * n
    return {"family": family, "num_points": num_points, "n_gp": n_gp, "cfg": cfg, "cells": n,
            "kernel_ms": ms, "wall_ms": wall * 1e3, "tflops": flops / (ms * 1e-3) / 1e12,
            "checks_per_s": n / (ms * 1e-3)}

runs = []
SWEEPS = [("cartpole", 48, 1024, 2), ("cartpole", 48, 1024, 1), ("pendulum", 512, 512, 2),
          ("pendulum", 512, 2048, 2), ("pendulum", 512, 2048, 1)]
if os.environ.get("SL_PROBE_SWEEPS") == "short":
    SWEEPS = SWEEPS[:1] + SWEEPS[3:4]
for args in SWEEPS:
    try:
        r = time_sweep(*args)
    except Exception as e:          # keep going: this is a probe
        r = {"args": args, "error": repr(e)}
    print(r, flush=True)
    runs.append(r)
out["sweeps"] = runs
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as f:
    json.dump(out, f, indent=1)
