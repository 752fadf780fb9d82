"""cProfile of update_safe_set inside the exploration loop (a data point added before every update).  Development probe."""
import os, sys, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_learning_amd as sl
from safe_learning_amd.benchmarks import build_lyapunov, table_case, _true_dynamics_numpy
n_gp = int(sys.argv[1]) if len(sys.argv) > 1 else 40
case = table_case(num_points=(2001, 1501), n_gp=n_gp, stack=True)
lyap = build_lyapunov(case)
warnings.simplefilter("ignore")
rng = np.random.default_rng(0)
def new_point():
    sa = np.concatenate([rng.uniform(-0.05, 0.05, 2), rng.uniform(-0.5, 0.5, 1)])[None, :]
    lyap.dynamics.add_data_point(sa, _true_dynamics_numpy(case, sa))
for _ in range(3):
    new_point(); lyap.update_safe_set()
torch.cuda.synchronize()
import time
times = []
pr = cProfile.Profile()
for _ in range(20):
    new_point()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable(); lyap.update_safe_set(); pr.disable()
    torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
print("update_safe_set after add_data_point: mean %.2f ms, min %.2f, max %.2f" % (np.mean(times), min(times), max(times)), [round(t, 2) for t in times])
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
