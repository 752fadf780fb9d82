#!/usr/bin/env python
"""One-GPU proxy of the multi-GPU run that no box of this pool can do.

A rank of the sharded ``update_safe_set`` (replacing ``lyapunov.py:512-606``) runs ``sl_lyap_sweep``
over its contiguous 64-aligned index range ``distributed.shard_bounds(n, N)[r : r + 2]`` with a
full replica of the model and nothing else of the other ranks: the shards of N = 2, 4, 8 are timed
here ONE AFTER ANOTHER on one MI355X (HIP events on the launch stream), exactly the launches the N
ranks would issue side by side.  What the proxy cannot see: the collectives (two 64-byte record
gathers + the mask-word gather per update - budgeted from the measured two-ranks-on-one-GPU /
RCCL-at-world-1 figures) and contention for nothing (the ranks share no resource on a node: one
GPU each, xGMI only for those small gathers).

Also counted per shard: the 16-cell runs of grid rows that cross a saturation kink of the policy
(`sl_gp4.hip`: a run with a kink restarts its k_x recurrence - such tiles cost ~2 % more; the
initial question of the review: do contiguous x_0 slabs hold them evenly?).

    python tools/shard_balance.py [--num-points 128] [--gpus 2 4 8] [--config C4|C5]

prints a markdown report (kept as profiles/r06_shard_balance.md) and, as its last line, a JSON
record.  With --weighted it also cuts the shards by the measured per-slab cost instead of equal
cell counts and times those (only worth it when the equal cut is more than 3 % off balance)."""

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kink_runs(case, bounds):
    """Number of 16-cell runs (consecutive cells of a row of the last grid axis) inside each shard in
    which the saturated linear policy u = clip(x K^T, -1, 1) has a kink."""
    num = np.asarray(case["num_points"])
    d = len(num)
    K = np.asarray(case["K"], dtype=float).reshape(-1)[:d]
    lo, hi = case["saturate"]
    axes = [np.linspace(l[0], l[1], n) for l, n in zip(case["limits"], num)]
    n_last = int(num[-1])
    rows = int(np.prod(num[:-1]))
    # u along a row is base + K_last x_last: its value at the first and last cell of every run
    lead = np.zeros(rows)
    stride = rows
    for k in range(d - 1):
        stride //= int(num[k])
        lead += np.tile(np.repeat(axes[k] * K[k], stride), rows // (stride * int(num[k])))
    starts = np.arange(0, n_last, 16)
    ends = np.minimum(starts + 15, n_last - 1)
    u0 = lead[:, None] + K[-1] * axes[-1][starts][None, :]
    u1 = lead[:, None] + K[-1] * axes[-1][ends][None, :]
    kink = (((u0 < lo) != (u1 < lo)) | ((u0 > hi) != (u1 > hi)))
    flat_start = (np.arange(rows)[:, None] * n_last + starts[None, :]).reshape(-1)
    kink = kink.reshape(-1)
    out = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        out.append(int(np.count_nonzero(kink[(flat_start >= a) & (flat_start < b)])))
    return out, int(kink.sum()), int(kink.size)


def time_ranges(fn, ranges, repeat):
    import torch
    times = []
    for lo, hi in ranges:
        best = []
        for _ in range(repeat):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(lo, hi)
            b.record()
            b.synchronize()
            best.append(a.elapsed_time(b))
        times.append(float(np.mean(best)))
    return times


def weighted_bounds(n, world, slab_cost, slab_cells, align=64):
    """Contiguous 64-aligned shards of (nearly) equal predicted cost from a per-slab cost profile."""
    cum = np.concatenate(([0.0], np.cumsum(slab_cost)))
    edges = np.concatenate(([0], np.cumsum(slab_cells)))
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        s = int(np.searchsorted(cum, target, side="right") - 1)
        s = min(max(s, 0), len(slab_cost) - 1)
        frac = (target - cum[s]) / max(slab_cost[s], 1e-300)
        cell = edges[s] + frac * slab_cells[s]
        cell = int(round(cell / align)) * align
        bounds.append(min(max(cell, bounds[-1]), n))
    bounds.append(n)
    return bounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4", choices=["C4", "C5"])
    ap.add_argument("--num-points", type=int, default=None)
    ap.add_argument("--n-gp", type=int, default=1024)
    ap.add_argument("--gpus", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--collective-ms", type=float, default=0.7,
                    help="budget per update for the collectives (profiles/r05_bench_two_ranks_one_gpu.log: "
                         "0.68 ms for the two record gathers + the mask gather between two ranks)")
    ap.add_argument("--weighted", action="store_true")
    args = ap.parse_args()
    import torch
    from safe_learning_amd import distributed as dist_utils
    from safe_learning_amd.benchmarks import build_lyapunov, headline_case
    torch.cuda.set_device(0)
    lines, record = [], {"config": args.config}
    if args.config == "C4":
        npts = args.num_points or 128
        case = headline_case(num_points=npts, n_gp=args.n_gp)
        lyap = build_lyapunov(case)
        lyap.update_safe_set()                           # uploads the model, allocates the seeds
        n = lyap.discretization.nindex
        ctx = lyap._ctx

        def sweep(lo, hi):
            ctx.lyap_sweep(lo, hi, lyap._d_init[lo // 64:], None if lyap._values_implicit else lyap._d_values[lo:],
                           lyap._d_neg[lo // 64:], lyap._d_result)

        unit, what = "cells", "Lyapunov.update_safe_set() sweep (k_gp_sweep4), cart-pole %d^4, %d-point GP" % (npts, args.n_gp)
        kink_fn = lambda b: kink_runs(case, b)           # noqa: E731
    else:
        sys.path.insert(0, ROOT)
        import bench
        npts = args.num_points or 64
        case = headline_case(num_points=npts, n_gp=args.n_gp)
        rl, actions = bench.build_policy_iteration(case)
        rl.successor_cache(0)                            # the recomputing sweep: what a rank's FIRST sweep costs
        rl.value_iteration(actions)
        n = rl.discretization.nindex
        ctx = rl._ctx
        v_new = torch.empty(n, dtype=torch.float64, device="cuda")
        arg = torch.empty(n, dtype=torch.int32, device="cuda")
        stats = torch.zeros(2, dtype=torch.float64, device="cuda")

        def sweep(lo, hi):
            ctx.bellman_sweep(lo, hi, actions, v_new, arg, None, stats)

        unit, what = "vertices", ("PolicyIteration.value_iteration(action_space) sweep, recomputing kernels "
                                  "(k_bellman4s + k_bellman_lookup), cart-pole %d^4 x 9 actions, %d-point GP" % (npts, args.n_gp))
        kink_fn = None
    sweep(0, n)
    full_ms = time_ranges(sweep, [(0, n)], max(args.repeat, 2))[0]
    lines.append("# Shard balance on one MI355X (round 6)\n")
    lines.append("Workload: %s.  Whole grid in one launch: **%.2f ms** (%d %s).\n" % (what, full_ms, n, unit))
    lines.append("Every shard `[lo, hi)` = `distributed.shard_bounds(n, N)` timed alone on the same GPU "
                 "(HIP events, %d launch(es) each); `kink runs` = 16-cell runs of grid rows that cross a "
                 "saturation kink of the policy.\n" % args.repeat)
    record.update(full_ms=full_ms, cells=int(n), shards={})
    slab_profile = None
    for world in args.gpus:
        bounds = dist_utils.shard_bounds(n, world)
        ranges = list(zip(bounds[:-1], bounds[1:]))
        ms = time_ranges(sweep, ranges, args.repeat)
        kinks = kink_fn(bounds)[0] if kink_fn else [None] * world
        live = [m for m, (a, b) in zip(ms, ranges) if b > a]
        balance = float(np.mean(live) / np.max(live))
        eff = full_ms / (world * max(live))
        rate = n / ((max(live) + args.collective_ms) * 1e-3)
        lines.append("## N = %d\n" % world)
        lines.append("| rank | range | %s | kernel ms | share of the slowest | kink runs |" % unit)
        lines.append("|---|---|---|---|---|---|")
        for r, ((a, b), m, k) in enumerate(zip(ranges, ms, kinks)):
            lines.append("| %d | [%d, %d) | %d | %.3f | %.4f | %s |" % (r, a, b, b - a, m, m / max(live), "-" if k is None else k))
        lines.append("")
        lines.append("balance (mean / max of the shards' kernel time): **%.4f**; sum of the shards %.2f ms "
                     "(whole grid %.2f ms); strong-scaling efficiency predicted from the slowest shard "
                     "`t_1 / (N max_r t_r)`: **%.4f**; predicted rate with %.2f ms of collectives per update: "
                     "**%.3e %s/s**\n" % (balance, sum(ms), full_ms, eff, args.collective_ms, rate, unit))
        record["shards"][str(world)] = {"bounds": [int(b) for b in bounds], "kernel_ms": ms, "kink_runs": kinks,
                                        "balance": balance, "efficiency": eff, "predicted_rate": rate}
        if world == max(args.gpus):
            slab_profile = (ms, [b - a for a, b in ranges])
    if args.weighted and slab_profile is not None:
        cost, cells = slab_profile
        lines.append("## Cost-weighted cuts (from the N = %d profile)\n" % max(args.gpus))
        for world in args.gpus:
            if world >= max(args.gpus):
                continue
            bounds = weighted_bounds(n, world, np.asarray(cost), np.asarray(cells))
            ranges = list(zip(bounds[:-1], bounds[1:]))
            ms = time_ranges(sweep, ranges, args.repeat)
            lines.append("N = %d: bounds %s, kernel ms %s, balance %.4f\n"
                         % (world, bounds, ["%.3f" % m for m in ms], float(np.mean(ms) / np.max(ms))))
            record.setdefault("weighted", {})[str(world)] = {"bounds": [int(b) for b in bounds], "kernel_ms": ms}
    print("\n".join(lines))
    print(json.dumps(record))


if __name__ == "__main__":
    main()
