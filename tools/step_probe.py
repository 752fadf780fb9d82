"""Where does the host time of one update_safe_set() go?  cProfile over many steps of a small
configuration (the step of C1 is bound by the host's enqueue time, not by its kernels)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    args = bench.parse_args(["--config", name]) if hasattr(bench, "parse_args") else None
    kind, label, case = bench.build_workload(args)
    from safe_learning_amd.benchmarks import build_lyapunov
    obj = build_lyapunov(case)
    for _ in range(20):
        obj.update_safe_set()
    torch.cuda.synchronize()
    for events in (False, True):
        obj._ctx.timing_configure(n if events else 0)
        t0 = time.perf_counter()
        for _ in range(n):
            obj.update_safe_set()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print("%s events=%s: host %.2f us/step, total %.2f us/step" % (name, events, 1e6 * host / n, 1e6 * total / n))
        obj._ctx.timing_configure(0)
    # what rounds 2-5 did: two torch.cuda.Event pairs per step, created and recorded by the host code
    keep = []
    t0 = time.perf_counter()
    for _ in range(n):
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
        pairs[0][0].record()
        pairs[1][0].record()
        obj.update_safe_set()
        pairs[0][1].record()
        pairs[1][1].record()
        keep.append(pairs)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("%s four torch events per step: host %.2f us/step, total %.2f us/step" % (name, 1e6 * host / n, 1e6 * total / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        obj.update_safe_set()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)


main()
