"""Which GP hyper-parameter sets give a non-degenerate headline workload?  (GPU, ~15 s per row)

    python tools/gp_variant_probe.py [num_points] > gpurun_out/variants.log

Prints, per variant, the number of cells that pass the decrease check, the size of the initial
set and of the safe set after one update_safe_set(), and c_max.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from safe_learning_amd.benchmarks import build_lyapunov, initial_safe_mask, make_case  # noqa: E402

ROWS = [
    dict(signal_std=0.05, noise_std=0.01, lengthscale=0.5, tau_scale=1.0),      # SURVEY 8d literal
    dict(signal_std=0.03, noise_std=0.0005, lengthscale=1.5, tau_scale=0.0),    # informed, tau 0
    dict(signal_std=0.03, noise_std=0.0005, lengthscale=1.5, tau_scale=0.0005),
    dict(signal_std=0.03, noise_std=0.0005, lengthscale=1.5, tau_scale=0.002),
    dict(signal_std=0.03, noise_std=0.0002, lengthscale=2.0, tau_scale=0.001),
    dict(signal_std=0.001, noise_std=0.0002, lengthscale=1.0, tau_scale=0.0),   # tight
    dict(signal_std=0.001, noise_std=0.0002, lengthscale=1.0, tau_scale=0.002),
    dict(signal_std=0.03, noise_std=0.0005, lengthscale=1.5, tau_scale=0.002, initial_radius=0.35),
]


def main():
    import torch
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    for kw in ROWS:
        t0 = time.time()
        case = make_case("cartpole", num_points=npts, n_gp=1024, **kw)
        lyap = build_lyapunov(case)
        lyap.update_safe_set()
        torch.cuda.synchronize()
        neg = int(torch.sum(_popcount(lyap._d_neg)).item())
        safe = int(torch.sum(_popcount(lyap._d_safe)).item())
        init = int(np.count_nonzero(initial_safe_mask(case)))
        print(json.dumps(dict(kw, num_points=npts, negative=neg, initial=init, safe=safe,
                              c_max=lyap.c_max, seconds=round(time.time() - t0, 1))), flush=True)
        del lyap


def _popcount(words):
    import torch
    x = words.clone()
    total = torch.zeros_like(x)
    for _ in range(64):
        total += x & 1
        x = (x >> 1) & 0x7FFFFFFFFFFFFFFF
    return total


if __name__ == "__main__":
    main()
