"""Headline benchmark: grid-cell Lyapunov checks per second (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--num-points 128] [--n-gp 1024]

Workload (BASELINE.json configs[3] / SURVEY 8d "C4"): cart-pole state grid 128^4 (2.68e8 cells),
1024-point shared-kernel RBF GP over [x, u] (p = 5 inputs, 4 outputs), quadratic LQR Lyapunov
function, saturated linear policy, per-dimension L_v = |2Px|.  A step is one full
``Lyapunov.update_safe_set()``: the fused GP posterior + decrease-check sweep over every cell, the
lexicographic-min reduction of the failing cell, and the streaming pass that writes the safe mask.
All inputs are resident in HBM (the model is uploaded before the timed region); data is synthetic.

With N > 1 (torchrun, one rank per GPU) the grid is sharded by contiguous index ranges: total
work is fixed, so scaling is "strong".
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6        # MI355X FP64 matrix (= vector) peak, SURVEY 8d / BASELINE.md


def flops_per_check(n, p, d_out):
    """Algorithmic FP64 flops per cell: kernel row + mean + triangular solve."""
    return n * (4 * p + 2) + 2 * n * d_out + (n * n + 2 * n)


def cpu_baseline(case, budget_s=20.0):
    """The oracle (NumPy float64 restatement of the reference's batch loop) timed on this host:
    whole 10 000-cell batches of the same grid / GP until ~budget_s of CPU time is spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    grid = olyap.discretization
    batch = 10000
    rng = np.random.default_rng(0)
    done, t0 = 0, time.perf_counter()
    while True:
        start = int(rng.integers(0, max(grid.nindex - batch, 1)))
        idx = np.arange(start, min(start + batch, grid.nindex))
        olyap.negative(grid.index_to_state(idx))
        done += len(idx)
        elapsed = time.perf_counter() - t0
        if elapsed > budget_s or done >= grid.nindex:
            break
    return {"value": done / elapsed, "unit": "checks/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d cells (%d random 10000-cell batches of the same grid and GP) in %.1f s; "
                      "NumPy/SciPy float64 oracle, BLAS threads = all cores" % (done, done // batch, elapsed)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--num-points", type=int, default=128)
    ap.add_argument("--n-gp", type=int, default=1024)
    ap.add_argument("--family", default="cartpole", choices=["cartpole", "pendulum"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)

    from safe_learning_amd.benchmarks import build_lyapunov, make_case
    case = make_case(args.family, num_points=args.num_points, n_gp=args.n_gp)
    lyap = build_lyapunov(case)                    # uploads the model, computes V on the grid
    ncells = lyap.discretization.nindex
    d, p, n_gp = case["d"], case["d"] + case["m"], args.n_gp

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lyap.update_safe_set()
    barrier()
    lyap.sweep_events = []                         # HIP events around the dominant kernel
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lyap.update_safe_set()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])

    kernel_ms = [a.elapsed_time(b) for a, b in lyap.sweep_events]
    cells_per_launch = lyap._hi - lyap._lo
    flops = flops_per_check(n_gp, p, d) * cells_per_launch
    avg_ms = float(np.mean(kernel_ms))
    achieved = flops / (avg_ms * 1e-3) / 1e12

    traffic = None
    try:                                           # PMC result of the same command (tools/pmc_traffic.py)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if world == 1 and args.num_points == 128 and args.n_gp == 1024 and args.family == "cartpole":
            traffic = pmc["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        out = {
            # BASELINE.json's metric string; `value` is the checks/sec, `ms_per_step` the
            # ms per safe_set update
            "metric": "grid-cell Lyapunov checks/sec + ms/safe_set-update, 4D 128^4 grid, 1k-pt GP",
            "value": ncells * args.steps / elapsed,
            "unit": "checks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s %d^%d GridWorld (%d cells), %d-point RBF GP dynamics, "
                                   "quadratic Lyapunov function, Lyapunov.update_safe_set()"
                                   % (args.family, args.num_points, d, ncells, n_gp),
                       "grid_sharding": "contiguous index ranges over %d GPU(s)" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": "k_gp_sweep", "kernel_ms": avg_ms,
                         "flops_per_check": flops_per_check(n_gp, p, d)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(case)
        print(json.dumps(out))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
