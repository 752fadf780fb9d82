import os, sys, warnings, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from safe_learning_amd.benchmarks import make_case
case = make_case("pendulum", num_points=101, n_gp=100)
rl, actions = bench.build_policy_iteration(case)
warnings.simplefilter("ignore")
for _ in range(5): rl.value_iteration(actions)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): rl.value_iteration(actions)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
