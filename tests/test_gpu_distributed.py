"""The sharded engine end to end with two ranks on ONE MI355X (gloo rendezvous on 127.0.0.1,
both ranks drive cuda:0): each rank sweeps its half of the grid with the HIP kernels, the key /
count / histogram reductions and the mask gather go through torch.distributed, and every rank
must reproduce the oracle's safe set.  (With one GPU per rank the only difference is the RCCL
backend.)"""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import exclusions
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results, backend="gloo", device_per_rank=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world == 1:
        os.environ["SL_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(rank % torch.cuda.device_count() if device_per_rank else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_lyapunov
    sl.config.gp_batch_size = oracle.config.gp_batch_size = 100
    failures = []
    scenarios = [
        ("det", cases.make_case("pendulum", num_points=50, dynamics="linear", tau_scale=0.02)),
        ("gp", cases.make_case("pendulum", num_points=45, n_gp=90, tau_scale=0.0)),
        ("cartpole-gp", cases.make_case("cartpole", num_points=7, n_gp=100, tau_scale=0.0)),
        # 50 cells over two ranks: the second shard is empty and starts at an unaligned index
        ("tiny", cases.make_case("1d", num_points=50)),
    ]
    for name, case in scenarios:
        lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
        assert lyap._world == world
        assert world == 1 or name == "tiny" or lyap._hi - lyap._lo < lyap.discretization.nindex
        # the constructor keeps only the shard of V; reading `values` is a LOCAL operation (the other
        # ranks' cells are computed here, no collective behind the attribute): rank 0 alone reads
        # it first - with a gather behind the property this line would wait for rank 1 forever
        if lyap._d_values_full is not None:
            failures.append((name, "V computed eagerly"))
        if rank == 0 and not np.array_equal(lyap.values, olyap.values):
            failures.append((name, "values, read on rank 0 only"))
        dist.barrier()
        if not np.array_equal(lyap.values, olyap.values):
            failures.append((name, "values"))
        if lyap._d_values_full is None or lyap._d_values_full.numel() != lyap.discretization.nindex:
            failures.append((name, "gathered V"))
        rng = np.random.default_rng(3)
        for step, shrink in enumerate((True, False, False)):
            if step == 1:
                extra = rng.choice(lyap.discretization.nindex,
                                   min(200, lyap.discretization.nindex // 4), replace=False)
                lyap.safe_set[extra] = True
                olyap.safe_set[extra] = True
                lyap.tau = olyap.tau = case["tau"] * 3 if case["tau"] else 0.0
            lyap.update_safe_set(can_shrink=shrink)
            olyap.update_safe_set(can_shrink=shrink)
            if not np.array_equal(lyap.safe_set, olyap.safe_set) or lyap.c_max != olyap.c_max:
                failures.append((name, step, int((lyap.safe_set != olyap.safe_set).sum()),
                                 lyap.c_max, olyap.c_max))
    # adaptive branch: shards swept by their owners, the sequential refinement pass replicated
    from safe_learning_amd.benchmarks import build_specs
    case = cases.make_case("pendulum", num_points=45, dynamics="linear", tau_scale=0.02)
    init = np.zeros(int(np.prod(case["num_points"])), dtype=bool)
    init[cases.initial_safe_mask(case)] = True
    policy, dynamics, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=init, adaptive=True)
    opolicy, odynamics, ovalue, olv = cases.oracle_specs(case)
    olyap = oracle.Lyapunov(oracle.GridWorld(case["limits"], case["num_points"]), ovalue, odynamics,
                            case["lf"], olv, case["tau"], opolicy, initial_set=init, adaptive=True)
    for shrink, factor in ((True, 1.0), (False, 1.5)):
        lyap.update_safe_set(can_shrink=shrink, max_refinement=16, safety_factor=factor)
        olyap.update_safe_set(can_shrink=shrink, max_refinement=16, safety_factor=factor)
        if (not np.array_equal(lyap.safe_set, olyap.safe_set) or lyap.c_max != olyap.c_max
                or not np.array_equal(lyap._refinement, olyap._refinement)):
            failures.append(("adaptive", shrink))
    # the refinement array is sharded like V: every rank keeps its cells; the shards were gathered
    # inside update_safe_set (which every rank calls), so a read on ONE rank is local too
    if lyap._refinement_dev is None or lyap._refinement_dev.numel() != lyap._hi - lyap._lo:
        failures.append(("adaptive", "refinement not sharded"))
    lyap._refinement_host = None
    if rank == world - 1 and not np.array_equal(lyap._refinement, olyap._refinement):
        failures.append(("adaptive", "refinement, read on the last rank only"))
    dist.barrier()
    # get_safe_sample on every rank (the mask words of the whole grid are on every rank): the same
    # pair and bound as the oracle's
    case = cases.make_case("pendulum", num_points=33, n_gp=40, tau_scale=0.01, signal_std=0.001,
                           noise_std=0.0002, lengthscale=1.0)
    lyap, olyap = build_lyapunov(case), cases.oracle_lyapunov(case)
    lyap.update_safe_set()
    olyap.update_safe_set()
    perturbations = np.array([[-0.2], [-0.05], [0.0], [0.05], [0.2]])
    limits = np.array([[-1.0, 1.0]])
    for positive in (True, False):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            pair, bound = sl.get_safe_sample(lyap, perturbations, limits, positive=positive)
            opair, obound = oracle.get_safe_sample(olyap, perturbations, limits, positive=positive)
        if not np.array_equal(pair, opair) or not np.isclose(bound, obound, rtol=1e-7, atol=0):
            failures.append(("sample", positive, pair.tolist(), opair.tolist(), bound, obound))
    # update_values(gather=True): the shards are gathered NOW, collectively, so that a later read on
    # one rank only is local (a lazy gather behind a rank guard would wait for the others forever)
    lyap.update_values(gather=True)
    if world > 1 and lyap._d_values_full is None:
        failures.append(("values", "not gathered eagerly"))
    if rank == 0 and not np.array_equal(lyap.values, olyap.values):
        failures.append(("values", "rank-guarded read"))
    # PolicyIteration across ranks: each rank sweeps its shard of the vertices (matrix-core
    # kernels; the shard boundary cuts a row of the last axis), the new table is all-gathered
    import test_gpu_rl
    case = cases.make_case("pendulum", num_points=[11, 65], n_gp=70)
    rl, orl, vf, ovf = test_gpu_rl._rl_pair(sl, case, [11, 65])
    assert rl._world == world
    actions = np.linspace(-1, 1, 9)[:, None]
    grid, ogrid = vf.discretization, ovf.discretization
    rl.policy = sl.Triangulation(grid, np.zeros((grid.nindex, 1)))
    orl.policy = oracle.Triangulation(ogrid, np.zeros((ogrid.nindex, 1)))
    rl.discrete_policy_optimization(actions)
    oq, _ = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok_q = np.ones_like(oq, dtype=bool)
    for a, action in enumerate(actions):
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        ok_q[:, a] = ~test_gpu_rl.ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
    if not exclusions.within("two ranks: discrete_policy_optimization", ok_q, "successor"):
        failures.append(("rl", "exclusions", float(1 - ok_q.mean())))
    top2 = np.sort(oq, axis=1)[:, -2:]
    tie = np.abs(top2[:, 1] - top2[:, 0]) <= 1e-9 * np.abs(top2[:, 1])
    best, obest = rl.policy._host_parameters()[:, 0], orl.policy.parameters[:, 0]
    if np.any((best != obest) & ok_q.all(axis=1) & ~tie):
        failures.append(("rl", "greedy policy"))
    rl.policy.parameters = orl.policy.parameters.copy()
    for sweep in range(2):
        vf.parameters = ovf.parameters.copy()
        nxt = orl.dynamics(x, orl.policy(x))
        nxt = nxt[0] if isinstance(nxt, tuple) else nxt
        res = rl.value_iteration()
        try:                                    # ambiguous vertices of the policy table: membership
            amb = exclusions.check_own_vertices("two ranks: value_iteration sweep %d" % sweep, orl,
                                                orl.policy, x, vf._host_parameters())
        except AssertionError as exc:
            failures.append(("rl", "membership", str(exc)))
            amb = test_gpu_rl.ambiguous_points(orl.policy, x)
        ok = ~amb & ~test_gpu_rl.ambiguous_points(ovf, nxt)
        if not exclusions.within("two ranks: value_iteration sweep %d" % sweep, ok | amb, "successor"):
            failures.append(("rl", "exclusions", float(1 - ok.mean())))
        orl.value_iteration()
        if not np.allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12):
            failures.append(("rl", "value_iteration", sweep))
        if not res >= 0.0:
            failures.append(("rl", "residual", res))
    # the 4x4x4 Bellman kernel (last axis = whole wavefronts) on the two shards
    case = cases.make_case("pendulum", num_points=[12, 64], n_gp=70)
    rl, orl, vf, ovf = test_gpu_rl._rl_pair(sl, case, [12, 64])
    rl.policy = sl.Triangulation(vf.discretization, np.zeros((vf.discretization.nindex, 1)))
    orl.policy = oracle.Triangulation(ovf.discretization, np.zeros((ovf.discretization.nindex, 1)))
    q = rl.discrete_policy_optimization(actions, return_values=True).cpu().numpy()
    oq, _ = orl.discrete_policy_optimization(actions)
    x = orl.state_space
    ok = np.ones(len(x), dtype=bool)
    for action in actions:
        nxt = orl.dynamics(x, np.broadcast_to(action, (len(x), 1)))
        ok &= ~test_gpu_rl.ambiguous_points(ovf, nxt[0] if isinstance(nxt, tuple) else nxt)
    if not exclusions.within("two ranks: 4x4x4 action values", ok, "successor"):
        failures.append(("rl", "exclusions", float(1 - ok.mean())))
    if q.shape != oq.shape or not np.allclose(q[ok], oq[ok], rtol=1e-9, atol=1e-12):
        failures.append(("rl", "4x4x4 action values"))
    # ... and the policy-evaluation sweep with the greedy table (k_bellman4_policy: six live rows
    # per rank in a workgroup step of eight)
    rl.policy.parameters = orl.policy.parameters.copy()
    vf.parameters = ovf.parameters.copy()
    nxt = orl.dynamics(x, orl.policy(x))
    rl.value_iteration()
    try:
        amb = exclusions.check_own_vertices("two ranks: 4x4x4 policy evaluation", orl, orl.policy, x,
                                            vf._host_parameters())
    except AssertionError as exc:
        failures.append(("rl", "membership", str(exc)))
        amb = test_gpu_rl.ambiguous_points(orl.policy, x)
    ok = ~amb & ~test_gpu_rl.ambiguous_points(ovf, nxt[0])
    if not exclusions.within("two ranks: 4x4x4 policy evaluation", ok | amb, "successor"):
        failures.append(("rl", "exclusions", float(1 - ok.mean())))
    orl.value_iteration()
    # (the greedy policy's successors were located by the max sweep above: served from this rank's
    # successor cache; k_bellman4_policy where a vertex's policy value is none of the cached actions)
    if not any(k in rl._ctx.last_kernel() for k in ("k_bellman_cached", "k_bellman4_policy")):
        failures.append(("rl", "policy kernel", rl._ctx.last_kernel()))
    if not np.allclose(vf._host_parameters()[ok], ovf.parameters[ok], rtol=1e-9, atol=1e-12):
        failures.append(("rl", "4x4x4 policy evaluation"))
    results[rank] = failures
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu():
    port = _free_port()
    results = mp.get_context("spawn").Manager().dict()    # never fork a process that holds a HIP context
    mp.spawn(_worker, args=(2, port, results), nprocs=2, join=True)
    for rank in range(2):
        assert results[rank] == [], results[rank]


def test_rccl_collectives_world_one():
    """The same scenarios over the `nccl` backend (RCCL) with one rank: every reduction, the key
    all-gather and the mask gather are issued as RCCL calls on device tensors."""
    port = _free_port()
    results = mp.get_context("spawn").Manager().dict()    # never fork a process that holds a HIP context
    mp.spawn(_worker, args=(1, port, results, "nccl"), nprocs=1, join=True)
    assert results[0] == [], results[0]


def test_rccl_one_gpu_per_rank():
    """The whole scenario list over RCCL with ONE GPU PER RANK - prefix rule, adaptive rule with its
    all-to-alls, the value-iteration all-gather, get_safe_sample.  Skipped on a one-GPU box; the
    driver's 8-GPU node runs it with min(8, devices) ranks, the first time RCCL moves a byte between
    two devices for this package."""
    devices = torch.cuda.device_count()
    if devices < 2:
        pytest.skip("needs at least two GPUs (found %d)" % devices)
    world = min(devices, 8)
    port = _free_port()
    results = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, results, "nccl", True), nprocs=world, join=True)
    for rank in range(world):
        assert results[rank] == [], (rank, results[rank])


def test_c_abi_rccl_collectives_one_gpu_per_rank():
    """sl_comm_init & co. between devices: the unique id of rank 0 handed to the others (the host
    application's job: here a manager dict)."""
    devices = torch.cuda.device_count()
    if devices < 2:
        pytest.skip("needs at least two GPUs (found %d)" % devices)
    world = min(devices, 8)
    manager = mp.get_context("spawn").Manager()
    results, shared = manager.dict(), manager.dict()
    mp.spawn(_abi_comm_worker, args=(world, results, shared), nprocs=world, join=True)
    for rank in range(world):
        assert results[rank] is True, rank


def _abi_comm_worker(rank, world, results, shared=None):
    """The RCCL entry points of the C ABI (no torch.distributed): communicator of `world` ranks."""
    import time
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(rank % torch.cuda.device_count() if world > 1 else 0)
    from safe_learning_amd import _hip
    ctx = _hip.Context()
    if world == 1:
        uid = _hip.Context.comm_unique_id()
    elif rank == 0:
        uid = _hip.Context.comm_unique_id()
        shared["uid"] = bytes(uid)
    else:
        while "uid" not in shared:
            time.sleep(0.05)
        uid = shared["uid"]
    assert len(uid) == 128
    ctx.comm_init(uid, rank, world)
    dev = ctx.torch_device
    # result record: folded in place (world 1: unchanged)
    rec = torch.tensor([5, 7, 3, 2, 9, 11, 4, 6], dtype=torch.int64, device=dev)
    ctx.allreduce_result(rec)
    hist = torch.arange(256, dtype=torch.int64, device=dev)
    ctx.allreduce_sum_u64(hist, 256)
    res = torch.tensor([0.25, 3.5], dtype=torch.float64, device=dev)
    ctx.allreduce_max_f64(res, 2)
    shard = torch.arange(1000, dtype=torch.float64, device=dev)
    full = torch.empty(1000 * world, dtype=torch.float64, device=dev)
    ctx.allgather(shard, full, shard.numel() * 8)
    ctx.synchronize()
    # (identical records on every rank: the keys fold to themselves, the two counters add up)
    ok = (rec.cpu().tolist() == [5, 7, 3, 2, 9, 11, 4 * world, 6 * world]
          and torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int64) * world)
          and res.cpu().tolist() == [0.25, 3.5]
          and torch.equal(full.cpu(), shard.cpu().repeat(world)))
    # errors: a second communicator on the same context, collectives after destroy
    try:
        ctx.comm_init(uid, rank, world)
        ok = False
    except _hip.HipEngineError:
        pass
    ctx.comm_destroy()
    try:
        ctx.allreduce_max_f64(res, 2)
        ok = False
    except _hip.HipEngineError:
        pass
    results[rank] = ok


def test_c_abi_rccl_collectives_world_one():
    """sl_comm_init / sl_allreduce_result / sl_allgather / sl_allreduce_* through RCCL itself."""
    results = mp.get_context("spawn").Manager().dict()
    mp.spawn(_abi_comm_worker, args=(1, results), nprocs=1, join=True)
    assert results[0] is True
