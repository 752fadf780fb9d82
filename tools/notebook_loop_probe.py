"""Wall-clock of the notebooks' exploration loop (examples/inverted_pendulum.ipynb cells 17-19:
update_safe_set -> get_safe_sample -> add_data_point) on the notebook's shape - 2001 x 1501 cells, table V
and table policy, a FunctionStack of two GPs - per phase.  Development probe, not a test.
    python tools/notebook_loop_probe.py [n_gp] [iterations] [num_samples]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import safe_learning_amd as sl
from safe_learning_amd.benchmarks import build_lyapunov, table_case, _true_dynamics_numpy

n_gp = int(sys.argv[1]) if len(sys.argv) > 1 else 40
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
num_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
case = table_case(num_points=(2001, 1501), n_gp=n_gp, stack=True)
lyap = build_lyapunov(case)
perturbations = np.array([[0.], [0.1], [-0.1], [0.2], [-0.2]])
limits = np.array([[-1., 1.]])
t = {"update_safe_set": [], "get_safe_sample": [], "add_data_point": []}


def clock(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    t[name].append(1e3 * (time.perf_counter() - t0))
    return out


with warnings.catch_warnings():
    warnings.simplefilter("ignore", RuntimeWarning)
    for it in range(iters):
        clock("update_safe_set", lyap.update_safe_set)
        np.random.seed(it)
        sa, bound = clock("get_safe_sample", lambda: sl.get_safe_sample(lyap, perturbations, limits,
                                                                         positive=True, num_samples=num_samples))
        y = _true_dynamics_numpy(case, sa)
        clock("add_data_point", lambda: lyap.dynamics.add_data_point(sa, y))
for k, v in t.items():
    print("%-16s first %.2f ms, then mean %.2f ms (min %.2f, max %.2f) over %d" % (k, v[0], np.mean(v[1:]), min(v[1:]), max(v[1:]), len(v) - 1))
print("safe cells", int(lyap.safe_set.sum()), "c_max", lyap.c_max, "kernel", lyap._ctx.last_kernel())
