"""Streaming pass at 128^4 (C4-lin's model): duration against the level - nothing below it (every span
cleared by its bound), the level of the real update, everything below it (no span cleared)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from safe_learning_amd import _hip  # noqa: E402


def main():
    args = bench.parse_args(["--config", "C4-lin"])
    kind, label, case = bench.build_workload(args)
    from safe_learning_amd.benchmarks import build_lyapunov
    lyap = build_lyapunov(case)
    lyap.update_safe_set()
    torch.cuda.synchronize()
    ctx, dev, n = lyap._ctx, lyap._ctx.torch_device, lyap.discretization.nindex
    real = lyap._d_folded.clone() if lyap._d_folded[_hip.R_FAIL_I] != 0 else lyap._d_result.clone()
    words = (n + 63) // 64
    safe = torch.zeros(words + 1, dtype=torch.int64, device=dev)
    record = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
    none_below = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
    none_below[_hip.R_FAIL_I] = -1
    all_below = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
    all_below[_hip.R_FAIL_V] = -1
    all_below[_hip.R_FAIL_I] = (1 << 63) - 1
    for name, folded in (("nothing below the level", none_below), ("the update's level", real), ("everything below", all_below)):
        for _ in range(3):
            ctx.lyap_finalize_dev(0, n, None, lyap._d_init, None, folded, None, safe, record)
        ctx.timing_configure(16)
        for _ in range(10):
            ctx.lyap_finalize_dev(0, n, None, lyap._d_init, None, folded, None, safe, record)
        ms = ctx.timing_collect(ctx.TIMING_FINALIZE)
        ctx.timing_configure(0)
        print("%-26s %.4f ms  (safe cells %d)" % (name, float(np.mean(ms)), int(record[_hip.R_SAFE])))


main()
