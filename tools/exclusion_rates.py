"""Rates at which the table-lookup parity tests exclude points (oracle side only, no GPU): the
successor states of every test shape of tests/test_gpu_rl.py and greedy / smooth policy tables read
at their own vertices - see tests/exclusions.py.  python tools/exclusion_rates.py"""
import sys, numpy as np, scipy.linalg
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases, oracle
from test_gpu_rl import ambiguous_points
def orl_pair(case, n_vgrid):
    d = case["d"]; limits = case["limits"]
    qmat = -scipy.linalg.block_diag(np.eye(d), 0.1 * np.eye(1))
    ovgrid = oracle.GridWorld(limits, n_vgrid)
    rng = np.random.default_rng(4)
    v0 = -rng.random((ovgrid.nindex, 1))
    opolicy, odynamics, _, _ = cases.oracle_specs(case)
    ovf = oracle.Triangulation(ovgrid, v0, project=True)
    orl = oracle.PolicyIteration(opolicy, odynamics, oracle.QuadraticFunction(qmat), ovf, gamma=0.95)
    return orl, ovf
def cat(ovf, pts, amb):
    disc=ovf.discretization
    lim=np.asarray(disc.limits)
    clipped=((pts<=lim[:,0])|(pts>=lim[:,1])).any(axis=1)
    unit = disc._center_states(pts, clip=True) % disc.unit_maxes
    rel = unit/disc.unit_maxes
    line=(np.minimum(rel,1-rel)<1e-9).any(axis=1)
    return dict(total=len(pts), amb=int(amb.sum()), amb_clipped=int((amb&clipped).sum()), amb_line=int((amb&line&~clipped).sum()), amb_other=int((amb&~line&~clipped).sum()), clipped=int(clipped.sum()), line=int(line.sum()))
P=[("pendulum", dict(dynamics="analytic"), 15, 9),
    ("pendulum", dict(n_gp=70), 15, 9),
    ("pendulum", dict(n_gp=70), [16, 32], 3),
    ("cartpole", dict(n_gp=90), 5, 9),
    ("cartpole", dict(n_gp=130), [3, 4, 4, 16], 16),
    ("cartpole", dict(n_gp=60, stack=True), 4, 9),
    ("pendulum", dict(n_gp=70, stack=True), [9, 65], 9),
    ("pendulum", dict(n_gp=70), [12, 64], 9),
    ("pendulum", dict(n_gp=70), [5, 128], 2),
    ("cartpole", dict(n_gp=90), [3, 4, 3, 64], 16),
    ("pendulum", dict(n_gp=70), [6, 101], 3),
    ("cartpole", dict(n_gp=90), [3, 3, 2, 70], 9),
    ("cartpole", dict(n_gp=130), [2, 3, 2, 128], 12),]
for name,kw,nv,na in P:
    case=cases.make_case(name,num_points=nv,**kw)
    orl,ovf=orl_pair(case,nv)
    x=orl.state_space
    actions=np.linspace(-1,1,na)[:,None]
    tot=dict()
    ok=np.ones(len(x),bool)
    for a in actions:
        nxt=orl.dynamics(x,np.broadcast_to(a,(len(x),1)))
        nxt=nxt[0] if isinstance(nxt,tuple) else nxt
        amb=ambiguous_points(ovf,nxt)
        ok&=~amb
        c=cat(ovf,nxt,amb)
        for k,v in c.items(): tot[k]=tot.get(k,0)+v
    print(name,kw,nv,na,'excluded vertices %.3f'%(1-ok.mean()), {k:round(v/tot['total'],3) for k,v in tot.items() if k!='total'})
print('--- policy tables evaluated at their own vertices')
for name,kw,nv,na in P[:5]+P[7:9]:
    case=cases.make_case(name,num_points=nv,**kw)
    orl,ovf=orl_pair(case,nv)
    x=orl.state_space
    actions=np.linspace(-1,1,na)[:,None]
    orl.policy=oracle.Triangulation(ovf.discretization,np.zeros((ovf.discretization.nindex,1)))
    oq,obest=orl.discrete_policy_optimization(actions)
    amb=ambiguous_points(orl.policy,x)
    u=orl.policy(x)
    tab=orl.policy.parameters
    print(name,kw,nv,na,'ambiguous vertices of the greedy table %.3f; interpolant != table at %.3f'%(amb.mean(), (u!=tab).mean()))
    table = np.linspace(-1, 1, len(x))[:, None]
    pol=oracle.Triangulation(ovf.discretization, table)
    print('    smooth table: %.3f'%ambiguous_points(pol,x).mean())
