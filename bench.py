"""Headline benchmark: grid-cell Lyapunov checks per second (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C4] [--num-points P] [--n-gp G]

Default workload (BASELINE.json configs[3] / SURVEY 8d "C4"): cart-pole state grid 128^4 (2.68e8
cells), 1024-point shared-kernel RBF GP over [x, u] (p = 5 inputs, 4 outputs), quadratic LQR
Lyapunov function, saturated linear policy, per-dimension L_v = |2Px|.  A step is one full
``Lyapunov.update_safe_set()``: the fused GP posterior + decrease-check sweep over every cell, the
lexicographic-min reduction of the failing cell, and the streaming pass that writes the safe mask.
All inputs are resident in HBM (the model is uploaded before the timed region); data is synthetic.

``--config`` selects the other BASELINE.json configurations (they are parity-test cases, not the
headline): C1 (1-D, 1001 cells), C2 (pendulum 256^2, 512-pt GP), C3 (pendulum 2048^2, 2048-pt GP,
LyapunovNetwork), C4-lin / C4-det (cart-pole 128^4 with linear / Euler dynamics, the HBM-bound
cases), C5 (cart-pole 64^4 x 9 actions: Bellman optimality sweeps, also run to convergence) and
C5-policy (the other sweep of that loop: evaluation of the greedy table policy).

``--gpus N``: one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank;
otherwise bench.py launches the N ranks itself (``torch.multiprocessing.spawn``), backend ``nccl``
(= RCCL over xGMI) when every rank has its own GPU, ``gloo`` when ranks have to share a device.
The grid is sharded by contiguous index ranges: total work is fixed, so scaling is "strong".
"""

import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6        # MI355X FP64 matrix (= vector) peak, SURVEY 8d / BASELINE.md
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
HEADLINE_METRIC = "grid-cell Lyapunov checks/sec + ms/safe_set-update, 4D 128^4 grid, 1k-pt GP"
CONFIGS = ("C1", "C2", "C2-table", "C2-table-large", "C2-table-stack", "C2-notebook", "C2-table-det", "C3", "C4", "C4-lin",
           "C4-det", "C5", "C5-policy")


def flops_per_check(n, p, d_out, heads=1):
    """Algorithmic FP64 flops per cell: kernel row + mean + triangular solve; a
    FunctionStack of `heads` single-output GPs repeats the first and third term per head."""
    return heads * (n * (4 * p + 2) + (n * n + 2 * n)) + 2 * n * d_out


# ---------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------
def build_workload(args):
    """-> (kind, label, case or RL tuple).  kind: 'lyapunov' or 'bellman'."""
    from safe_learning_amd.benchmarks import GP_VARIANTS, headline_case, make_case, network_weights
    cfg = args.config
    informed = GP_VARIANTS["informed"]
    if cfg == "C4":
        npts, n_gp = args.num_points or 128, args.n_gp or 1024
        case = headline_case(num_points=npts, n_gp=n_gp, family=args.family, variant=args.gp_variant)
        if args.gp_variant == "survey":      # SURVEY 8d's literal inputs: full tau as well
            case = headline_case(num_points=npts, n_gp=n_gp, family=args.family, variant="survey")
            case["tau"] = float(np.sum(2.0 / (np.asarray(case["num_points"]) - 1)) / 2.0)
        label = ("%s %d^%d GridWorld (%d cells), %d-point RBF GP dynamics, quadratic Lyapunov "
                 "function, Lyapunov.update_safe_set()"
                 % (args.family, npts, case["d"], npts ** case["d"], n_gp))
        if args.gp_variant:
            label += " [GP hyper-parameters: %s]" % args.gp_variant
    elif cfg == "C1":
        case = make_case("1d", num_points=args.num_points or 1001)
        label = "1-D GridWorld (%d cells), linear dynamics, quadratic V" % case["num_points"][0]
    elif cfg == "C2":
        npts, n_gp = args.num_points or 256, args.n_gp or 512
        case = make_case("pendulum", num_points=npts, n_gp=n_gp, tau_scale=0.01, **informed)
        label = "pendulum %d^2 GridWorld, %d-point RBF GP dynamics, quadratic V" % (npts, n_gp)
    elif cfg in ("C2-table", "C2-table-large", "C2-table-stack", "C2-notebook", "C2-table-det"):
        from safe_learning_amd.benchmarks import notebook_kernels, table_case
        shape = (args.num_points,) * 2 if args.num_points else (
            (251, 251) if cfg == "C2-table" else (2001, 1501))
        det = cfg == "C2-table-det"
        # C2-table-stack: the notebooks' dynamics model proper - a FunctionStack of one
        # single-output GP per state dimension (inverted_pendulum.ipynb:152-181)
        case = table_case(num_points=shape, n_gp=args.n_gp or 128,
                          dynamics="analytic" if det else None,
                          stack=cfg in ("C2-table-stack", "C2-notebook"))
        if cfg == "C2-notebook":     # ... with the notebook's kernels: Linear + Matern32 * Linear
            case["dynamics"]["kernels"] = notebook_kernels(case)
        label = ("pendulum %dx%d GridWorld, %s dynamics, V and policy piecewise-linear "
                 "tables on 101x101 vertices, L_v = |grad V| (inverted_pendulum.ipynb cell 14)"
                 % (shape[0], shape[1], "explicit-Euler pendulum" if det
                    else ("FunctionStack of two %d-point Linear + Matern32 x Linear GPs"
                          if cfg == "C2-notebook" else
                          "FunctionStack of two %d-point RBF GPs" if cfg == "C2-table-stack"
                          else "%d-point RBF GP") % (args.n_gp or 128)))
    elif cfg == "C3":
        npts, n_gp = args.num_points or 2048, args.n_gp or 2048
        case = make_case("pendulum", num_points=npts, n_gp=n_gp, tau_scale=0.0, **informed)
        case["V"] = {"kind": "network", "layer_dims": [64, 64, 64], "activations": ["tanh"] * 3,
                     "eps": 1e-8, "weights": network_weights(2, [64, 64, 64], seed=1)}
        case["lv"] = ("norm_grad",)
        label = ("pendulum %d^2 GridWorld, %d-point RBF GP dynamics, LyapunovNetwork [64,64,64], "
                 "L_v = |grad V|_1" % (npts, n_gp))
    elif cfg in ("C4-lin", "C4-det"):
        npts = args.num_points or 128
        dyn = "linear" if cfg == "C4-lin" else "analytic"
        case = make_case("cartpole", num_points=npts, dynamics=dyn,
                         tau_scale=0.004 if cfg == "C4-lin" else 0.0)
        label = "cartpole %d^4 GridWorld, %s dynamics, quadratic V" % (
            npts, "linear" if cfg == "C4-lin" else "10-step explicit-Euler cart-pole")
    elif cfg == "C5":
        npts, n_gp = args.num_points or 64, args.n_gp or 1024
        case = headline_case(num_points=npts, n_gp=n_gp)
        label = ("cartpole %d^4 value table x 9 actions, %d-point RBF GP mean dynamics, "
                 "PolicyIteration.value_iteration(action_space) Bellman sweeps" % (npts, n_gp))
        return "bellman", label, case
    elif cfg == "C5-policy":
        # the other sweep of the loop: evaluation of the greedy table policy (the policy that
        # discrete_policy_optimization leaves after three max sweeps from V = 0)
        npts, n_gp = args.num_points or 64, args.n_gp or 1024
        case = headline_case(num_points=npts, n_gp=n_gp)
        label = ("cartpole %d^4 value table, %d-point RBF GP mean dynamics, greedy 9-action table "
                 "policy, PolicyIteration.value_iteration() policy-evaluation sweeps" % (npts, n_gp))
        return "policy", label, case
    else:
        raise ValueError(cfg)
    return "lyapunov", label, case


def build_policy_iteration(case):
    import scipy.linalg
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs
    policy, dynamics, _, _ = build_specs(case)
    grid = sl.GridWorld(case["limits"], case["num_points"])
    vf = sl.Triangulation(grid, np.zeros((grid.nindex, 1)), project=True)
    d = case["d"]
    reward = sl.QuadraticFunction(-scipy.linalg.block_diag(0.1 * np.eye(d), 0.1 * np.eye(1)))
    rl = sl.PolicyIteration(policy, dynamics, reward, vf, gamma=0.98)
    return rl, np.linspace(-1, 1, 9)[:, None]


# ---------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle on this host's cores, bounded samples
# ---------------------------------------------------------------------------------------------
def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas, threads = "unknown", os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        libs = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if libs:
            blas = "%s %s (%s threads)" % (libs[0].get("internal_api"), libs[0].get("version"),
                                           libs[0].get("num_threads"))
            threads = int(libs[0].get("num_threads") or threads)
    except Exception:
        pass
    return model, blas, threads


def _oracle_batches(case, budget_s, threads=None, min_cells=0):
    """Whole 10 000-cell batches (lyapunov.py:517-529) of the same grid / model at random places
    until ~budget_s of wall time is spent and at least min_cells cells are done -> (cells, seconds)."""
    import cases
    from threadpoolctl import threadpool_limits
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    grid = olyap.discretization
    batch = 10000
    rng = np.random.default_rng(0)
    with threadpool_limits(limits=threads):
        olyap.negative(grid.index_to_state(np.arange(min(batch, grid.nindex))))     # warm-up
        done, t0 = 0, time.perf_counter()
        while True:
            start = int(rng.integers(0, max(grid.nindex - batch, 1)))
            idx = np.arange(start, min(start + batch, grid.nindex))
            olyap.negative(grid.index_to_state(idx))
            done += len(idx)
            elapsed = time.perf_counter() - t0
            if (elapsed > budget_s and done >= min_cells) or done >= 4 * grid.nindex:
                break
    return done, elapsed


def _reference_faithful(case, max_cells=6_000_000):
    """What a user of the reference experiences: ``update_safe_set`` with the global ``np.argsort``
    and the early exit at the first failing cell (lyapunov.py:512-587), through the oracle.  The
    whole grid when it has at most ``max_cells`` cells, otherwise the central slab of the first
    axis (the layers around x_0 = 0, which contain the initial safe set)."""
    import copy
    import cases
    num = list(case["num_points"])
    total = int(np.prod(num))
    sub = copy.copy(case)
    sample = "whole grid (%d cells)" % total
    if total > max_cells:
        rest = int(np.prod(num[1:]))
        layers = max(2, min(num[0], max_cells // rest))
        lo_i = (num[0] - layers) // 2
        axis = np.linspace(case["limits"][0][0], case["limits"][0][1], num[0])
        sub["limits"] = [[float(axis[lo_i]), float(axis[lo_i + layers - 1])]] + list(case["limits"][1:])
        sub["num_points"] = [layers] + num[1:]
        sample = ("central slab of %d of the %d layers of the first axis (%d of %d cells), same "
                  "cell spacing and model" % (layers, num[0], layers * rest, total))
    t0 = time.perf_counter()
    olyap = cases.oracle_lyapunov(sub)                       # all_points + V on every cell
    t_values = time.perf_counter() - t0
    t0 = time.perf_counter()
    olyap.update_safe_set()                                  # argsort + batches + early exit
    t_update = time.perf_counter() - t0
    return {"reference_faithful_ms": 1e3 * t_update, "reference_faithful_values_ms": 1e3 * t_values,
            "reference_faithful_sample": sample,
            "reference_faithful_safe_cells": int(olyap.safe_set.sum())}


def cpu_baseline(kind, case, budget_s=14.0, min_cells=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    model, blas, threads = _cpu_info()
    out = {"unit": "checks/s", "cores": threads, "kind": "port", "cpu_model": model,
           "blas": blas, "host_cpus": os.cpu_count()}
    if kind == "bellman":
        import cases
        import oracle
        import scipy.linalg
        # one Jacobi sweep x 9 actions of the oracle on a sample of vertices
        policy, dynamics, _, _ = cases.oracle_specs(case)
        grid = oracle.GridWorld(case["limits"], case["num_points"])
        vf = oracle.Triangulation(grid, np.zeros((grid.nindex, 1)), project=True)
        d = case["d"]
        reward = oracle.QuadraticFunction(-scipy.linalg.block_diag(0.1 * np.eye(d), 0.1 * np.eye(1)))
        orl = oracle.PolicyIteration(policy, dynamics, reward, vf, gamma=0.98)
        rng = np.random.default_rng(0)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            idx = rng.integers(0, grid.nindex, 20000)
            x = grid.index_to_state(idx)
            for a in np.linspace(-1, 1, 9):
                orl.future_values(x, actions=np.full((len(x), 1), a))
            done += len(idx) * 9
        elapsed = time.perf_counter() - t0
        out.update(value=done / elapsed, unit="(vertex, action) pairs/s",
                   sample="%d (vertex, action) pairs (random 20000-vertex batches x 9 actions) in "
                          "%.1f s; NumPy/SciPy float64 oracle, BLAS threads = all cores"
                          % (done, elapsed))
        return out
    done, elapsed = _oracle_batches(case, budget_s)
    done4, elapsed4 = _oracle_batches(case, budget_s / 2, threads=4, min_cells=min_cells)
    out["default_threads_value"] = done / elapsed
    out["threads4_value"] = done4 / elapsed4       # the notebooks run with num_cores = 4
    # `value` is the faster of the two thread settings (small batches do not scale across a big
    # socket: 4 threads beat the BLAS default on the 256-CPU host), `cores` the threads it used
    if done4 / elapsed4 > done / elapsed:
        done, elapsed, out["cores"] = done4, elapsed4, 4
    out["value"] = done / elapsed
    out["sample"] = ("%d cells (%d random 10000-cell batches of the same grid and model) in %.1f s "
                     "with %d BLAS threads; NumPy/SciPy float64 oracle"
                     % (done, max(done // 10000, 1), elapsed, out["cores"]))
    out.update(_reference_faithful(case))
    # "an OpenMP C++ build of the same loop if NumPy is > 10x off" - it is (87 GFLOP/s of a
    # 2 x 64-core host).  The NumPy figures above stay as what a user of the reference experiences
    # (its notebooks run TensorFlow with 4 threads); `value` becomes what the host cores can do.
    try:
        strong = _all_cores_baseline(case, min_cells)
    except (ValueError, RuntimeError, OSError) as exc:     # configurations cpu_sweep.cpp does not restate
        out["all_cores_note"] = "oracle/cpu_sweep.cpp not used: %s" % exc
        return out
    out["numpy_value"], out["numpy_cores"], out["numpy_sample"] = out["value"], out["cores"], out["sample"]
    out["numpy_label"] = "reference user's experience (NumPy/SciPy oracle, 4 BLAS threads like the notebooks)"
    out.update(value=strong["value"], cores=strong["threads"], sample=strong["sample"], all_cores=strong)
    return out


def _cpu_quota():
    """CPUs' worth of time the cgroup of this process may use per period (None: unlimited)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def _all_cores_baseline(case, min_cells=0, budget_s=60.0):
    """The batch loop of lyapunov.py:517-529 as C++ / OpenMP over all the cores this process may use
    (oracle/cpu_sweep.cpp; measurement only - the NumPy oracle stays the checker and checks one of
    the batches here): random 10 000-cell batches of the same grid and model, at least
    max(min_cells, 2^24) cells when that fits `budget_s` at the calibrated rate."""
    import cases
    from oracle import cpu_sweep
    olyap = cases.oracle_lyapunov(case, compute_values=False)
    gp = getattr(olyap.dynamics, "gaussian_process", None)
    if gp is None:
        raise ValueError("no shared-kernel GP dynamics")
    sweep = cpu_sweep.CpuSweep(case, gp)
    n, batch = olyap.discretization.nindex, 10000
    rng = np.random.default_rng(0)

    def batches(count):
        starts = rng.integers(0, max(n - batch, 1), count)
        return np.concatenate([np.arange(s, min(s + batch, n)) for s in starts])

    # the cores this process may really use: its affinity mask, cut by the cgroup's CPU quota (a
    # container with 256 visible CPUs and a quota of 32 runs 128 OpenMP threads at a quarter speed)
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cpu_quota()
    nthreads = max(1, min(allowed, int(quota + 0.999)) if quota else allowed)
    first = batches(1)
    neg, rec, _, threads = sweep.check(first, threads=nthreads, records=True)   # warm-up + the checked batch
    ref = cases.oracle_cell_records(olyap, first)
    scale = float(np.abs(ref[:, 0]).max())
    diff = float(np.abs(rec[:, 0] - ref[:, 0]).max())
    flips = int(np.count_nonzero(neg != (ref[:, 0] < ref[:, 1])))
    # the thread count that is fastest HERE (SMT siblings, a CPU quota the cgroup files do not show,
    # other tenants of the host): calibrated on a small sample
    calib = batches(max(2 * nthreads * 32 // batch, 8))
    rate, tried = 0.0, {}
    for cand in sorted({nthreads, max(1, nthreads // 2), max(1, nthreads // 4)}, reverse=True):
        while True:                                   # at least 0.3 s per candidate
            _, _, sec, _ = sweep.check(calib, threads=cand)
            if sec >= 0.3 or len(calib) >= min(n, 1 << 22):
                break
            calib = batches(4 * len(calib) // batch)
        tried[cand] = len(calib) / sec
        if tried[cand] > rate:
            rate, best = tried[cand], cand
    nthreads = best
    want = max(int(min_cells), min(n, 1 << 24))
    cells = int(min(want, max(rate * budget_s, len(calib))))
    idx = batches(max(-(-cells // batch), 1))
    cpu0 = time.process_time()
    _, _, sec, threads = sweep.check(idx, threads=nthreads)
    busy = (time.process_time() - cpu0) / sec                               # CPUs kept busy on average
    value = len(idx) / sec
    return {"value": value, "threads": threads, "cells": int(len(idx)), "seconds": sec,
            "cpus_in_affinity_mask": allowed, "cgroup_cpu_quota": quota, "cpus_busy_on_average": busy,
            "calibration_checks_per_s_by_threads": {str(k): v for k, v in tried.items()},
            "gflops": value * sweep.flops_per_check / 1e9, "kind": "port",
            "checked": {"cells": int(len(first)), "max_abs_decrease_difference": diff, "decrease_scale": scale,
                        "mask_flips": flips, "against": "NumPy oracle (oracle/np_lyapunov.py), same batch"},
            "sample": ("%d cells (%d random 10000-cell batches of the same grid and model) in %.2f s on %d "
                       "OpenMP threads (%.1f CPUs busy on average): oracle/cpu_sweep.cpp, the batch loop of lyapunov.py:517-529 in C++ "
                       "(%.0f GFLOP/s of SURVEY 8d's %d flops per check)%s"
                       % (len(idx), len(idx) // batch, sec, threads, busy, value * sweep.flops_per_check / 1e9,
                          sweep.flops_per_check,
                          "" if len(idx) >= want else "; fewer than the %d cells asked for: bounded by %.0f s" % (want, budget_s)))}


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
def run_rank(args, rank, world, local_rank, backend):
    import torch
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(ndev, 1))
    dist = None
    # SL_FORCE_COLLECTIVES=1: the exact collective sequence of the N > 1 path at world size 1
    # (RCCL on a one-GPU box, tests/test_gpu_bench.py)
    grouped = world > 1 or os.environ.get("SL_FORCE_COLLECTIVES") == "1"
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29511")
        backend = backend or "nccl"
        dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    kind, label, case = build_workload(args)
    extra = {}
    if kind == "lyapunov":
        from safe_learning_amd.benchmarks import build_lyapunov
        obj = build_lyapunov(case)                # uploads the model, computes V on the grid
        units = obj.discretization.nindex
        step = obj.update_safe_set
        cells_per_launch = obj._hi - obj._lo
    elif kind == "policy":
        obj, actions = build_policy_iteration(case)
        if args.successor_cache == "off":
            obj.successor_cache(0)
        for _ in range(3):
            obj.value_iteration(actions)
        obj.discrete_policy_optimization(actions)
        units = obj.discretization.nindex
        step = obj.value_iteration
        cells_per_launch = obj._hi - obj._lo
    else:
        obj, actions = build_policy_iteration(case)
        units = obj.discretization.nindex * len(actions)
        step = lambda: obj.value_iteration(actions)          # noqa: E731
        cells_per_launch = obj._hi - obj._lo
        if args.successor_cache == "off":
            obj.successor_cache(0)
        else:
            # (i) the recomputing sweep - what every sweep cost before the successor cache and what the
            # first sweep of a loop still costs: a few sweeps with the cache switched off, HIP
            # events around the kernels; (ii) the loop as a user runs it, from V = 0: the first
            # sweep locates the successors and fills the cache, the timed steps below and the
            # convergence run behind them are served from it
            obj.successor_cache(0)
            obj.value_iteration(actions)
            barrier()
            obj._ctx.timing_configure(8)
            t0 = time.perf_counter()
            for _ in range(3):
                obj.value_iteration(actions)
            barrier()
            un_ms = 1e3 * (time.perf_counter() - t0) / 3
            un_kernel_ms = float(np.mean(obj._ctx.timing_collect(obj._ctx.TIMING_BELLMAN)))
            obj._ctx.timing_configure(0)
            extra["uncached_sweep"] = {"ms_per_sweep": un_ms, "kernel_ms": un_kernel_ms,
                                       "kernel": obj._ctx.last_kernel()}
            obj.successor_cache(-1)
            obj.value_function.parameters = np.zeros((obj.discretization.nindex, 1))
            barrier()
            loop_t0 = time.perf_counter()
            obj.value_iteration(actions)
            barrier()
            extra["first_sweep_ms"] = 1e3 * (time.perf_counter() - loop_t0)
            extra["first_sweep_kernel"] = obj._ctx.last_kernel()

    from safe_learning_amd import distributed as dist_utils
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the dominant kernel and around the streaming pass (k_finalize_dev), recorded
    # by the library on the stream it launches on (sl_timing_configure: events made here, not per step)
    obj._ctx.timing_configure(4 * args.steps + 16)
    dist_utils.start_timing()                     # device events around every collective
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    own_elapsed = time.perf_counter() - t0        # this rank's host clock before the barrier
    barrier()
    elapsed = time.perf_counter() - t0
    collective_ms = dist_utils.stop_timing() / max(args.steps, 1)
    kernel_ms = obj._ctx.timing_collect(obj._ctx.TIMING_LYAP_SWEEP if kind == "lyapunov" else obj._ctx.TIMING_BELLMAN)
    finalize_ms = obj._ctx.timing_collect(obj._ctx.TIMING_FINALIZE)
    obj._ctx.timing_configure(0)
    avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    per_rank = [elapsed, avg_ms, collective_ms, 1e3 * own_elapsed / max(args.steps, 1)]
    if grouped:
        t = torch.tensor(per_rank, dtype=torch.float64, device="cuda")
        gathered = torch.empty(world * t.numel(), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(gathered, t)
        per_rank_all = gathered.reshape(world, -1).cpu().tolist()
        elapsed = max(r[0] for r in per_rank_all)
        observed_world = dist.get_world_size()
        # what every rank actually ran: its kernel, its device, its share of the grid
        mine = {"rank": rank, "kernel": obj._ctx.last_kernel(), "cells": int(cells_per_launch),
                "device": int(torch.cuda.current_device()), "range": [int(obj._lo), int(obj._hi)]}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
    else:
        per_rank_all, observed_world = [per_rank], 1
        ranks_info = [{"rank": 0, "kernel": obj._ctx.last_kernel(), "cells": int(cells_per_launch),
                       "device": int(torch.cuda.current_device()), "range": [int(obj._lo), int(obj._hi)]}]
    # --gpus N has to be what ran: N ranks, the grid cut into N shards, the same kernel on each
    if observed_world != world or (args.gpus and world != args.gpus and os.environ.get("WORLD_SIZE")):
        raise SystemExit("bench.py: --gpus %d but %d rank(s) in the process group (WORLD_SIZE=%s)"
                         % (args.gpus, observed_world, os.environ.get("WORLD_SIZE")))
    if sum(r["cells"] for r in ranks_info) != obj.discretization.nindex:
        raise SystemExit("bench.py: the ranks' shards hold %d cells, the grid has %d"
                         % (sum(r["cells"] for r in ranks_info), obj.discretization.nindex))
    if len({r["kernel"].split("<")[0] for r in ranks_info if r["cells"] > 0}) > 1:
        raise SystemExit("bench.py: the ranks ran different kernels: %r" % [r["kernel"] for r in ranks_info])

    # ---- everything below is outside the timed region --------------------------------------
    if kind == "lyapunov":
        from safe_learning_amd.benchmarks import initial_safe_mask
        extra["safe_cells"] = int(obj.safe_count)
        extra["initial_cells"] = int(np.count_nonzero(initial_safe_mask(case)))
        extra["c_max"] = float(obj.c_max)
        extra["tau"] = float(case["tau"])
        if "dynamics" in case and case["dynamics"].get("kind") == "gp":
            dyn = case["dynamics"]
            extra["gp_hyper"] = {"signal_std": float(np.sqrt(dyn["variance"])),
                                 "noise_std": float(np.sqrt(dyn["noise_variance"])),
                                 "lengthscale": float(np.ravel(dyn["lengthscales"])[0]),
                                 "variant": args.gp_variant or "informed",
                                 "note": "default = the 'informed' set (not SURVEY 8d's literal 0.05 / 0.01 / 0.5 "
                                         "with the full tau, under which one cell of 2.7e8 passes): same cost per "
                                         "cell, a mask that is not vacuous; --gp-variant survey runs the literal set"}
        # cells that pass the decrease check (popcount of the mask words of all ranks)
        neg = obj._d_neg[:(cells_per_launch + 63) // 64]
        cnt = torch.tensor([int(_popcount(neg))], dtype=torch.int64, device="cuda")
        if grouped:
            dist.all_reduce(cnt)
        extra["negative_cells"] = int(cnt[0])
        # update_safe_set() returns when its kernels are enqueued; the reference's call also leaves
        # c_max on the host (lyapunov.py:590-595).  The same loop with c_max READ after every update
        # (the 64-byte record copy + the host code behind it - and, when nothing failed, the select
        # passes of the no-failure quirk - inside the clock): two steps, labelled.
        barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            obj.update_safe_set()
            obj.c_max
        barrier()
        extra["ms_per_step_incl_c_max_read"] = 1e3 * (time.perf_counter() - t1) / 2
        # end to end: one more update including the bool[N] mask on the host (lyapunov.py:598-606)
        barrier()
        t1 = time.perf_counter()
        obj.update_safe_set()
        mask = obj.safe_set
        end_to_end_ms = 1e3 * (time.perf_counter() - t1)
        assert int(mask.sum()) == extra["safe_cells"]
        if args.config in ("C2", "C4") and extra["safe_cells"] <= extra["initial_cells"] \
                and not args.diagnostic and args.gp_variant != "survey":
            # (attribution runs skip phases; SURVEY 8d's literal hyper-parameters are known to
            # let one cell of 2.7e8 pass - that variant exists to show the cost is the same)
            raise SystemExit("bench.py: degenerate workload - the level set did not grow (%d safe "
                             "cells, %d initial)" % (extra["safe_cells"], extra["initial_cells"]))
    elif kind == "policy":
        end_to_end_ms = None
        table = obj.policy._host_parameters().reshape(-1, case["num_points"][-1])
        extra["distinct_actions_per_row"] = float(
            np.mean((np.diff(np.sort(table, axis=1), axis=1) != 0).sum(axis=1) + 1))
    else:
        end_to_end_ms = None
        # sweeps to convergence at max|dV| <= 1e-6 max|V| (SURVEY 8d metric iii), continuing from
        # the table the timed sweeps left
        sweeps, rel, last, monotone = args.warmup + args.steps, float("inf"), float("inf"), True
        if args.successor_cache != "off":
            sweeps += 1                               # the filling sweep
        while sweeps < args.max_sweeps:
            res = obj.value_iteration(actions)
            sweeps += 1
            vmax = float(obj.value_function._device_table.abs().max())
            rel = res / max(vmax, 1e-300)
            monotone = monotone and res <= last * (1 + 1e-12)
            last = res
            if rel <= 1e-6:
                break
        extra.update(sweeps_to_convergence=sweeps, relative_residual=rel,
                     residual_monotone=bool(monotone), converged=bool(rel <= 1e-6))
        if args.successor_cache != "off":
            # wall clock of the whole loop from V = 0: filling sweep + warm-up + timed steps + the rest
            torch.cuda.synchronize()
            extra["time_to_convergence_s"] = time.perf_counter() - loop_t0
            extra["steady_sweep_ms"] = 1e3 * elapsed / args.steps
            extra["successor_cache"] = obj.successor_cache_info

    if rank == 0:
        d = case["d"]
        dyn = case.get("dynamics", {})
        out = {
            "metric": HEADLINE_METRIC if args.config == "C4" else
            ("Bellman sweep (vertex, action) pairs/sec + ms/sweep" if kind == "bellman" else
             "policy-evaluation sweep vertices/sec + ms/sweep" if kind == "policy" else
             "grid-cell Lyapunov checks/sec + ms/safe_set-update"),
            "value": units * args.steps / elapsed,
            "unit": "checks/s" if kind == "lyapunov" else
            ("vertices/s" if kind == "policy" else "(vertex, action) pairs/s"),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": dict({"workload": label, "name": args.config,
                            "grid_sharding": "contiguous 64-aligned index ranges over %d GPU(s)" % world,
                            "collectives": {"backend": backend if grouped else None,
                                            "world_size": observed_world,
                                            "ranks_share_a_device": bool(world > max(ndev, 1))},
                            # per step; collective_ms = device time inside the collectives of a
                            # step (two 64-byte record gathers + the mask-word gather for
                            # update_safe_set; table gather + residual for a Bellman sweep),
                            # including the wait for the slowest rank to arrive; max over ranks
                            "collective_ms": max(r[2] for r in per_rank_all),
                            "per_rank_collective_ms": [r[2] for r in per_rank_all],
                            "per_rank_ms_per_step": [r[3] for r in per_rank_all],
                            "per_rank_kernel_ms": [r[1] for r in per_rank_all],
                            "per_rank_cells": [r["cells"] for r in ranks_info],
                            "per_rank_kernel": [r["kernel"].split("(")[0].strip() for r in ranks_info],
                            "per_rank_device": [r["device"] for r in ranks_info]}, **extra),
        }
        out["config"]["env_switches"] = env_switches()
        diag = {k: os.environ[k] for k in REFUSED_ENV if os.environ.get(k)}
        if diag:                                   # only reachable with --diagnostic
            out["diagnostic"] = diag
            out["metric"] = "DIAGNOSTIC (not a benchmark result): " + out["metric"]
        if end_to_end_ms is not None:
            out["end_to_end_ms"] = end_to_end_ms      # incl. bits->bytes and the bool[N] D2H
        out["roofline"] = roofline(args, kind, case, dyn, d, cells_per_launch, avg_ms, world)
        if "uncached_sweep" in extra:              # the recomputing sweep against ITS roof (the GEMM)
            un = extra["uncached_sweep"]
            flops = 2.0 * len(dyn["X"]) * 9 * d
            un["roofline"] = {"bound": "mfma", "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "achieved": flops * cells_per_launch / (un["kernel_ms"] * 1e-3) / 1e12}
            un["roofline"]["frac"] = un["roofline"]["achieved"] / FP64_MFMA_PEAK_TFLOPS
        if finalize_ms:
            fin = float(np.mean(finalize_ms)) * len(finalize_ms) / max(args.steps, 1)   # per step
            out["roofline"]["finalize_ms"] = fin
            out["roofline"]["step_kernels_ms"] = avg_ms + fin
            if "bytes_per_check" in out["roofline"]:       # (GB/s of the whole step, analytic dynamics)
                out["roofline"]["step_achieved"] = (out["roofline"]["bytes_per_check"] * cells_per_launch
                                                    / ((avg_ms + fin) * 1e-3) / 1e9)
                out["roofline"]["step_frac"] = out["roofline"]["step_achieved"] / HBM_PEAK_GBPS
                out["roofline"]["values_implicit"] = bool(getattr(obj, "_values_implicit", False))
        out["roofline"]["kernel"] = obj._ctx.last_kernel()    # what the library launched (sl_last_kernel)
        if out["roofline"]["kernel"].startswith("k_gp_small") and out["roofline"]["bound"] == "mfma":
            # <= 256 training points: as many FP64 exponentials as GEMM flops per cell - the vector
            # ALU's issue slots are the roof (profiles/pmc_valu.json), the matrix-pipe share stays beside it
            out["roofline"] = valu_roof(args, out["roofline"])
        # a build whose code audit failed compiles the 4x4x4 kernels out with only a warning
        # (safe_learning_amd/_build.py): the headline lines must not silently run on the fallbacks
        expected = {"C4": "k_gp_sweep4", "C3": "k_gp_sweep4", "C5": "k_bellman4"}.get(args.config)
        forced = any(os.environ.get(k) for k in ("SL_GP_CFG", "SL_BELLMAN4", "SL_BELLMAN_MFMA"))
        forced = forced or args.num_points or args.n_gp      # other shapes may pick other kernels
        if args.config == "C5" and args.successor_cache != "off" and not forced:
            # the timed steps come from the cache; the recomputing sweep in front of them must have
            # run on the 4x4x4 kernels
            if not (out["roofline"]["kernel"].startswith("k_bellman_cached")
                    and extra["uncached_sweep"]["kernel"].startswith(expected)
                    and "k_bellman_lookup" in extra["first_sweep_kernel"]):
                raise SystemExit("bench.py: C5 ran on %r / %r / %r" % (
                    out["roofline"]["kernel"], extra["uncached_sweep"]["kernel"], extra["first_sweep_kernel"]))
            expected = None
        if expected and not forced and not out["roofline"]["kernel"].startswith(expected):
            raise SystemExit("bench.py: %s ran on %r instead of %s* (library built without its "
                             "4x4x4 kernels?)" % (args.config, out["roofline"]["kernel"], expected))
        if world == 1 and not args.no_cpu_baseline and kind != "policy":
            # (the policy-evaluation line has no CPU leg of its own: C5's is the same oracle sweep)
            out["cpu_baseline"] = cpu_baseline(kind, case, min_cells=args.cpu_cells)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def _popcount(words):
    """Number of set bits in an int64 tensor (device side)."""
    import torch
    x = words.view(torch.uint8)
    table = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=x.device)
    return table[x.long()].sum()


def roofline(args, kind, case, dyn, d, cells_per_launch, avg_ms, world):
    """Achieved rate of the dominant kernel against the roofline that bounds it."""
    is_gp = dyn.get("kind") == "gp"
    if kind == "policy" and args.successor_cache != "off":
        # the vertex's own action and the vertex itself from the successor cache: two entries of
        # 8 d + 5 bytes, the policy value (8) and its action index (1), old and new value (16)
        bytes_per_vertex = 2 * (8.0 * d + 5) + 25
        achieved = bytes_per_vertex * cells_per_launch / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": None,
                "kernel_ms": avg_ms, "bytes_per_vertex": bytes_per_vertex,
                "gathered_bytes_per_vertex": 2 * (d + 1) * 8.0}
    if kind == "policy":
        n = len(dyn["X"])
        flops = 2.0 * n * d                      # the mean of the vertex's own action
        achieved = flops * cells_per_launch / (avg_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": None, "kernel_ms": avg_ms, "flops_per_vertex": flops,
                "note": "algorithmic flops = the posterior mean of each vertex's own action; the "
                        "kernel computes the means of every distinct action of a 64-cell tile for all "
                        "of its cells (one quarter-block GEMM per distinct action, DESIGN 4.4)"}
    if kind == "bellman" and args.successor_cache != "off":
        # sweeps served from the successor cache (sl_succ.hip) stream it: per (vertex, action) pair
        # 8 d weights + 4 (corner) + 1 (simplex) bytes, per vertex 8 (old value) + 8 (new value) +
        # 4 (arg-max); the (d + 1) gathered table values per pair come out of L2 / Infinity Cache
        # (134 MB table; SURVEY 8d counts them as cache resident)
        bytes_per_vertex = 9 * (8.0 * d + 5) + 20
        achieved = bytes_per_vertex * cells_per_launch / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": None,
                "kernel_ms": avg_ms, "bytes_per_vertex": bytes_per_vertex,
                "gathered_bytes_per_vertex": 9 * (d + 1) * 8.0}
    if kind == "bellman":
        n = len(dyn["X"])
        flops = 2.0 * n * 9 * d                  # one FP64 GEMM [cells x n] . [n x A*D]
        achieved = flops * cells_per_launch / (avg_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": None,            # filled in from sl_last_kernel by the caller
                "kernel_ms": avg_ms, "flops_per_vertex": flops}
    if is_gp:
        n, p = len(dyn["X"]), d + case["m"]
        heads = d if case.get("stack") else 1
        fpc = flops_per_check(n, p, d, heads)
        achieved = fpc * cells_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, source = None, None
        try:                                       # PMC passes of the same command (tools/pmc_traffic.py)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if world == 1 and args.config == "C4" and not args.num_points and not args.n_gp:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_traffic
                if pmc.get("source_sha256") == pmc_traffic.kernel_source_sha():
                    traffic, source = pmc["bytes_per_launch"], "profiles/pmc_traffic.json (rocprofv3 " \
                        "FETCH_SIZE x2 + WRITE_SIZE of this command on this version of sl_gp4.hip, " \
                        "separate passes, tools/profile_r06.sh; not this run)"
                else:
                    source = "profiles/pmc_traffic.json was measured on another version of sl_gp4.hip: " \
                        "re-run tools/profile_r06.sh"
        except (OSError, ValueError, KeyError):
            pass
        return {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": source, "kernel": None, "kernel_ms": avg_ms,
                "flops_per_check": fpc,
                "note": "compute-bound: 8.25 algorithmic HBM bytes per check (SURVEY 8d)"}
    # analytic dynamics: SURVEY 8d's 10 B per check (V read 8 + init mask + mask write, states
    # generated from the index); the kernel moves 8 B + 2 bits
    bytes_per_check = 10.0
    achieved = bytes_per_check * cells_per_launch / (avg_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": None,
           "kernel_ms": avg_ms, "bytes_per_check": bytes_per_check,
           "note": "the bit-exactness contract (one rounding per multiply and per add, no FMA) keeps the "
                   "FP64-VALU floor above the byte floor: ~95 operations per cart-pole cell with linear "
                   "dynamics (k_det_rows shares the row prefixes of the ordered sums; ~165 per cell when "
                   "every cell starts from scratch, ~600 with the Euler dynamics) against 10 bytes"}
    return valu_roof(args, out)


def valu_roof(args, out):
    """The deterministic sweeps and k_gp_small are bound by vector-ALU ISSUE, not by bytes or the
    matrix pipe: report that roof - the measured issue utilisation of the dominant kernel, from the
    PMC passes of this command (profiles/pmc_valu.json, tools/pmc_valu.py; tied to the kernel
    sources by a sha256, a stale entry is not reported) - and keep the byte / matrix-pipe figure
    as `<bound>_frac` beside it."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_valu.json")) as f:
            entry = json.load(f).get(args.config)
    except (OSError, ValueError):
        entry = None
    if not entry or args.num_points or args.n_gp:
        return out
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_valu
    if entry.get("source_sha256") != pmc_valu.sources_sha(entry["kernel"]):
        out["valu_note"] = "profiles/pmc_valu.json[%s] was measured on other kernel sources: re-run " \
                           "tools/profile_r06.sh" % args.config
        return out
    old_bound = out["bound"]
    out[old_bound + "_achieved"], out[old_bound + "_frac"] = out["achieved"], out["frac"]
    out[old_bound + "_peak"], out[old_bound + "_unit"] = out["peak"], out["unit"]
    out.update(bound="valu", achieved=entry["valu_issue_utilisation"], peak=1.0, frac=entry["valu_issue_utilisation"],
               unit="share of the SIMDs' cycles issuing a vector-ALU instruction "
                    "(SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024))",
               valu_kernel=entry["kernel"],
               valu_source="profiles/pmc_valu.json (rocprofv3 counter passes of this command on these kernel "
                           "sources, tools/profile_r06.sh; not this run)")
    for key in ("fp64_share_of_valu_instructions", "mfma_busy", "waves_waiting"):
        if key in entry:
            out[key] = entry[key]
    return out


# ---------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned(rank, args, world, port, backend):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run_rank(args, rank, world, rank, backend)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4", choices=CONFIGS)
    ap.add_argument("--num-points", type=int, default=None)
    ap.add_argument("--n-gp", type=int, default=None)
    ap.add_argument("--family", default="cartpole", choices=["cartpole", "pendulum"])
    ap.add_argument("--max-sweeps", type=int, default=3000, help="C5: bound of the convergence run")
    ap.add_argument("--successor-cache", default="on", choices=["on", "off"],
                    help="C5 / C5-policy: serve the sweeps after the first from the successor cache "
                         "(default) or recompute every sweep (the kernels of the first sweep)")
    ap.add_argument("--diagnostic", action="store_true",
                    help="allow SL_LIB_PATH / SL_GP4_SKIP / SL_BM_FLAGS / SL_B4P_FLAGS (development "
                         "builds, timing attribution): the line is tagged 'diagnostic' and its metric "
                         "renamed - not a benchmark result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=0,
                    help="least number of cells of the CPU baseline sample (SURVEY 8d asks for 2^24 = "
                         "16777216 at C4: ~3.5 minutes on 4 BLAS threads; the default sample is bounded "
                         "by time so that the whole run stays within minutes)")
    ap.add_argument("--gp-variant", default=None, choices=["survey", "informed", "tight"],
                    help="C4: GP hyper-parameter set (default: informed); 'survey' = SURVEY 8d's literal "
                         "inputs incl. the full tau, kept to show that the cost does not depend on them")
    ap.add_argument("--backend", default=None,
                    help="torch.distributed backend; default nccl (= RCCL), gloo if ranks share a GPU")
    return ap.parse_args(argv)


# Environment variables under which a bench line is not what the shipped library does: a development
# library (SL_LIB_PATH, tools/build_variant.sh) may skip work (SL_GP4_SKIP, SL_BM_FLAGS, SL_B4P_FLAGS -
# the shipped library does not read them).  The A/B switches of the shipped kernels are legitimate
# runs but not the default path: they are listed in the line (config.env_switches).
REFUSED_ENV = ("SL_LIB_PATH", "SL_GP4_SKIP", "SL_BM_FLAGS", "SL_B4P_FLAGS", "SL_GPS_FLAGS")
AB_ENV = ("SL_GP_CFG", "SL_GP_SMALL", "SL_GP_SMALL_WAVES", "SL_GP_SMALL_SPLIT", "SL_DET_ROWS", "SL_GP4_ONE_PANEL", "SL_GP4_SEEDS",
          "SL_GP4_TICKETS", "SL_BELLMAN_MFMA", "SL_BELLMAN4", "SL_BELLMAN4_POLICY", "SL_BELLMAN4_POLICY_CACHE",
          "SL_BELLMAN4_RAGGED", "SL_BELLMAN4_QUARTER", "SL_BELLMAN4_SPLIT", "SL_BELLMAN4_ROUND",
          "SL_BELLMAN4_SHARED", "SL_SUCC_CACHE", "SL_FORCE_COLLECTIVES")


def env_switches():
    return {k: os.environ[k] for k in AB_ENV if os.environ.get(k)}


def main():
    args = parse_args()
    bad = [k for k in REFUSED_ENV if os.environ.get(k)]
    if bad and not args.diagnostic:
        raise SystemExit("bench.py: %s set - a development library / work-skipping switches do not "
                         "produce benchmark lines (--diagnostic runs them, tagged as such)" % ", ".join(bad))
    import torch
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:                     # launched by torchrun: this process is one rank
        world = int(env_world)
        ndev = torch.cuda.device_count()
        backend = args.backend or ("nccl" if ndev >= min(world, args.gpus or world) else "gloo")
        run_rank(args, int(os.environ.get("RANK", "0")), world,
                 int(os.environ.get("LOCAL_RANK", "0")), backend)
        return
    world = max(1, args.gpus)
    if world == 1:
        run_rank(args, 0, 1, 0, args.backend)
        return
    # self-launch: one process per GPU (never fork a process that may hold a HIP context)
    import torch.multiprocessing as mp
    ndev = torch.cuda.device_count()
    backend = args.backend or ("nccl" if ndev >= world else "gloo")
    mp.spawn(_spawned, args=(args, world, _free_port(), backend), nprocs=world, join=True)


if __name__ == "__main__":
    main()
