"""Wall-clock per call of a policy-iteration loop on the notebooks' RL shape (a 2-D pendulum value table of a few
thousand vertices, a handful of actions): value_iteration(action_space), discrete_policy_optimization, value_iteration()
(policy evaluation), bellmann_error.  Development probe.   python tools/rl_loop_probe.py [points per axis] [n_gp | 0 = analytic]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_learning_amd.benchmarks import make_case
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 101
n_gp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
case = make_case("pendulum", num_points=npts, n_gp=n_gp or None, dynamics=None if n_gp else "analytic")
rl, actions = bench.build_policy_iteration(case)
warnings.simplefilter("ignore")


def clock(fn, reps=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


print("%d^2 vertices, %s dynamics" % (npts, ("%d-point GP" % n_gp) if n_gp else "analytic"))
print("value_iteration(action_space)      %.3f ms  [%s]" % (clock(lambda: rl.value_iteration(actions)), rl._ctx.last_kernel()))
print("discrete_policy_optimization       %.3f ms  [%s]" % (clock(lambda: rl.discrete_policy_optimization(actions)), rl._ctx.last_kernel()))
print("value_iteration() (evaluation)     %.3f ms  [%s]" % (clock(lambda: rl.value_iteration()), rl._ctx.last_kernel()))
print("bellmann_error()                   %.3f ms  [%s]" % (clock(lambda: rl.bellmann_error()), rl._ctx.last_kernel()))
