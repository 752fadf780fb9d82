import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ['SL_GP_CFG']='2'
import numpy as np, torch
import cases
from gp_cases import INFORMED, TIGHT
from safe_learning_amd.benchmarks import build_lyapunov
from test_gpu_lyapunov import _engine_records, _oracle_all
for name, kw in [("pendulum", dict(num_points=16, n_gp=600, tau_scale=0.01, **INFORMED)), ("cartpole", dict(num_points=4, n_gp=520, tau_scale=0.0, **TIGHT))]:
    case = cases.make_case(name, **kw)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    values, neg, rec = _engine_records(lyap)
    ref_rec, ref_neg = _oracle_all(olyap)
    d=case['d']
    # GP correction = mean - prior
    x = olyap.discretization.index_to_state(np.arange(len(rec)))
    u = olyap.policy(x)
    prior = np.hstack((x,u)) @ case['dynamics']['prior'].T
    got = rec[:,2:2+d]-prior; ref = ref_rec[:,2:2+d]-prior
    print(name, 'max abs err', np.abs(got-ref).max(0))
    np.set_printoptions(precision=4, linewidth=200, suppress=False)
    print('cells 0..19 got\n', got[:20].T, '\nref\n', ref[:20].T)
    print('ratio', (got[:20]/ref[:20]).T)
