import os, sys, warnings, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import safe_learning_amd as sl
from safe_learning_amd.benchmarks import build_lyapunov, table_case
case = table_case(num_points=(2001, 1501), n_gp=40, stack=True)
lyap = build_lyapunov(case)
pert = np.array([[0.], [0.1], [-0.1], [0.2], [-0.2]]); limits = np.array([[-1., 1.]])
warnings.simplefilter("ignore")
lyap.update_safe_set()
for _ in range(3): sl.get_safe_sample(lyap, pert, limits, positive=True, num_samples=1000)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): sl.get_safe_sample(lyap, pert, limits, positive=True, num_samples=1000)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(35)
