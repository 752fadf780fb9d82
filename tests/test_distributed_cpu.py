"""The N > 1 path on CPU: world_size 2 and 3 over gloo (127.0.0.1).

Every rank owns a contiguous 64-aligned shard, runs ``prefix_rule`` (the production
orchestration of safe_learning_amd.lyapunov) with the NumPy shard engine standing in for the HIP
kernels, and the assembled mask / c_max must equal the oracle's sequential result."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle
from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scenarios():
    out = []
    c = cases.make_case("pendulum", num_points=50, dynamics="linear", tau_scale=0.02)
    out.append(("shrink", c, [dict(can_shrink=True)]))
    out.append(("no_shrink", c, [dict(can_shrink=True), dict(can_shrink=False, extra=300, tau=4.0),
                                 dict(can_shrink=False, tau=0.0)]))
    c = cases.make_case("pendulum", num_points=24, dynamics="linear", tau_scale=0.01)
    c["P"] = np.array([[1.0, 0.0], [0.0, 0.0]])
    c["lv"] = ("const", 0.05)
    out.append(("ties", c, [dict(can_shrink=True), dict(can_shrink=False, tau=3.0)]))
    c = cases.make_case("pendulum", num_points=31, dynamics="linear", tau_scale=0.0)
    c["K"] = c["K"] * 0.0
    c["saturate"] = None
    c["dynamics"] = {"kind": "linear", "matrix": np.hstack((0.5 * np.eye(2), np.zeros((2, 1))))}
    c["initial_radius"] = 0.05
    out.append(("all_safe", c, [dict(can_shrink=True)]))
    c = cases.make_case("pendulum", num_points=20, dynamics="linear", tau_scale=50.0)
    c["initial_radius"] = -1.0
    out.append(("first_fails", c, [dict(can_shrink=True)]))
    c = cases.make_case("cartpole", num_points=7, dynamics="analytic", tau_scale=0.0)
    out.append(("cartpole", c, [dict(can_shrink=True), dict(can_shrink=False, extra=50)]))
    # 150 cells: with 8 ranks the shards are [0,64) [64,128) [128,150) and five EMPTY tail shards
    # that start at the unaligned index 150 - the shape of a 128^4-like split with spare ranks
    c = cases.make_case("1d", num_points=150)
    out.append(("tail", c, [dict(can_shrink=True), dict(can_shrink=False, extra=20, tau=2.0)]))
    return out


def _worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from np_shard_engine import NumpyShardEngine
    from safe_learning_amd import distributed as du
    from safe_learning_amd.lyapunov import prefix_rule
    oracle.config.gp_batch_size = 100
    device = torch.device("cpu")
    failures = []
    for name, case, steps in _scenarios():
        olyap = cases.oracle_lyapunov(case)
        n = olyap.discretization.nindex
        init = np.zeros(n, dtype=bool)
        init[cases.initial_safe_mask(case)] = True
        lo, hi = du.shard_range(n)
        assert lo % 64 == 0 or lo == n
        prev = init.copy()
        rng = np.random.default_rng(7)
        for step in steps:
            if "tau" in step:
                olyap.tau = case["tau"] * step["tau"]
            if "extra" in step:
                extra = rng.choice(n, step["extra"], replace=False)
                olyap.safe_set[extra] = True
                prev[extra] = True
            negative = olyap.negative(olyap.discretization.index_to_state(np.arange(n)))
            engine = NumpyShardEngine(lo, hi, olyap.values, negative, init, prev)
            c_max = prefix_rule(engine, n, 100, step["can_shrink"])
            bounds = du.shard_bounds(n, world)
            sizes = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
            full = du.allgather_concat(torch.from_numpy(engine.safe.astype(np.uint8)), sizes)
            safe = full.numpy().astype(bool)
            # the production form: every rank's buffer has the capacity of a full shard, ONE
            # all_gather_into_tensor into a pre-sized buffer, cut at n (Lyapunov._d_safe & co.)
            padded = torch.zeros(max(bounds[1] - bounds[0], 1), dtype=torch.uint8)
            padded[:hi - lo] = torch.from_numpy(engine.safe.astype(np.uint8))
            if not torch.equal(du.allgather_equal(padded, n), full):
                failures.append((name, "allgather_equal"))
            olyap.update_safe_set(can_shrink=step["can_shrink"])
            same_c = (c_max == olyap.c_max) or (np.isnan(c_max) and np.isnan(olyap.c_max))
            if not np.array_equal(safe, olyap.safe_set) or not same_c:
                failures.append((name, step, int((safe != olyap.safe_set).sum()), c_max, olyap.c_max))
            prev = olyap.safe_set.copy()
    results[rank] = failures
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_prefix_rule_gloo(world):
    port = _free_port()
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for rank in range(world):
        assert results[rank] == [], results[rank]


def test_shard_bounds():
    from safe_learning_amd.distributed import shard_bounds
    for n in (1, 63, 64, 65, 1000, 128 ** 4):
        for world in (1, 2, 3, 8):
            b = shard_bounds(n, world)
            assert b[0] == 0 and b[-1] == n and len(b) == world + 1
            assert all(x <= y for x, y in zip(b, b[1:]))
            assert all(x % 64 == 0 for x in b[1:-1] if x != n)


def test_single_process_matches_oracle_semantics():
    """world = 1 without torch.distributed: same engine, same result."""
    from np_shard_engine import NumpyShardEngine
    from safe_learning_amd.lyapunov import prefix_rule
    old = oracle.config.gp_batch_size
    oracle.config.gp_batch_size = 100
    try:
        for name, case, steps in _scenarios():
            olyap = cases.oracle_lyapunov(case)
            n = olyap.discretization.nindex
            init = np.zeros(n, dtype=bool)
            init[cases.initial_safe_mask(case)] = True
            negative = olyap.negative(olyap.discretization.index_to_state(np.arange(n)))
            engine = NumpyShardEngine(0, n, olyap.values, negative, init, init)
            c_max = prefix_rule(engine, n, 100, True)
            olyap.update_safe_set()
            assert np.array_equal(engine.safe, olyap.safe_set), name
            assert c_max == olyap.c_max, name
    finally:
        oracle.config.gp_batch_size = old


def _gather_worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from safe_learning_amd import distributed as du
    ok = True
    # contiguous-shard pattern (equal, one shorter, empty) and an arbitrary pattern
    for sizes in ([5, 5, 3, 0][:world] if world == 4 else [4] * (world - 1) + [1], [2, 0, 7, 1][:world]):
        sizes = list(sizes) + [0] * (world - len(sizes))
        mine = torch.arange(sizes[rank], dtype=torch.float64) + 100.0 * rank
        full = du.allgather_concat(mine, sizes)
        want = torch.cat([torch.arange(s, dtype=torch.float64) + 100.0 * r for r, s in enumerate(sizes)])
        ok = ok and torch.equal(full, want)
    # a reused receive buffer is really reused
    buf = torch.empty(world * 3, dtype=torch.int64)
    out = du.allgather_equal(torch.full((3,), rank, dtype=torch.int64), out=buf)
    ok = ok and out.data_ptr() == buf.data_ptr() and out.tolist() == [r for r in range(world) for _ in range(3)]
    # collective timing: spans are recorded and summed
    du.start_timing()
    du.allreduce_sum_(torch.ones(4))
    records, count = du.gather_records(torch.arange(8, dtype=torch.int64) + rank)
    ok = ok and count == world and records.reshape(world, 8)[:, 0].tolist() == list(range(world))
    ms = du.stop_timing()
    ok = ok and ms > 0.0 and du._timing is None
    results[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_single_tensor_gathers_gloo():
    port = _free_port()
    results = mp.Manager().dict()
    mp.spawn(_gather_worker, args=(4, port, results), nprocs=4, join=True)
    assert all(results[r] for r in range(4)), dict(results)


# ---- the adaptive branch (lyapunov.py:445-487, 540-582) across ranks --------------------------------
def _adaptive_scenarios():
    out = []
    for tau_scale, refinement, factor, batch in ((0.1, 3, 1.0, 60), (0.03, 6, 1.4, 60), (0.003, 3, 1.0, 25),
                                                 (0.003, 6, 1.4, 200)):
        case = cases.make_case("pendulum", num_points=25, dynamics="analytic", tau_scale=tau_scale)
        case["limits"] = [[-1.0, 1.03], [-0.97, 1.0]]
        out.append((case, refinement, factor, batch))
    case = cases.make_case("cartpole", num_points=6, dynamics="analytic", tau_scale=0.02)
    out.append((case, 5, 1.2, 100))
    return out


def _adaptive_run(rank, world):
    """adaptive_rule on the NumPy engine of this rank's shard against the oracle's sequential loop."""
    from np_shard_engine import NumpyAdaptiveEngine, NumpyShardEngine
    from safe_learning_amd import distributed as du
    from safe_learning_amd.lyapunov import adaptive_rule
    failures = []
    for case, refinement, factor, batch in _adaptive_scenarios():
        oracle.config.gp_batch_size = batch
        olyap = cases.oracle_lyapunov(case)
        olyap.adaptive = True
        n = olyap.discretization.nindex
        init = np.zeros(n, dtype=bool)
        init[cases.initial_safe_mask(case)] = True
        states = olyap.discretization.index_to_state(np.arange(n))
        nxt = olyap.dynamics(states, olyap.policy(states))
        decrease = olyap.v_decrease_bound(states, nxt)[:, 0]
        thr0 = np.broadcast_to(olyap.threshold(states, 1.0), (n, 1))[:, 0]
        bounds = du.shard_bounds(n, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        prior_safe = prior_ref = None
        for shrink in (True, False, False):
            shard = NumpyShardEngine(lo, hi, olyap.values, np.zeros(n, dtype=bool), init, init)
            engine = NumpyAdaptiveEngine(lo, hi, olyap.values, decrease, thr0, olyap.tau, init,
                                         None if shrink else prior_safe, None if shrink else prior_ref, shard)
            stats = {}
            c_max = adaptive_rule(engine, n, batch, refinement, factor, stats)
            olyap.update_safe_set(can_shrink=shrink, max_refinement=refinement, safety_factor=factor)
            sizes = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
            if world > 1:
                safe = du.allgather_concat(torch.from_numpy(engine.safe.astype(np.uint8)), sizes).numpy().astype(bool)
                ref = du.allgather_concat(torch.from_numpy(engine.refinement), sizes).numpy()
            else:
                safe, ref = engine.safe, engine.refinement
            same_c = c_max == olyap.c_max or (np.isnan(c_max) and np.isnan(olyap.c_max))
            if (not np.array_equal(safe, olyap.safe_set) or not np.array_equal(ref, olyap._refinement)
                    or not same_c or stats["safe"] != int(olyap.safe_set.sum())):
                failures.append((case["name"], refinement, batch, shrink, int((safe != olyap.safe_set).sum()),
                                 int((ref != olyap._refinement).sum()), c_max, olyap.c_max))
            prior_safe, prior_ref = olyap.safe_set.copy(), np.asarray(olyap._refinement).copy()
            olyap.tau = olyap.tau * 0.5          # the next call sees other thresholds
    return failures


def _adaptive_worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    results[rank] = _adaptive_run(rank, world)
    dist.barrier()
    dist.destroy_process_group()


def test_adaptive_rule_single_process():
    old = oracle.config.gp_batch_size
    try:
        assert _adaptive_run(0, 1) == []
    finally:
        oracle.config.gp_batch_size = old


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adaptive_rule_gloo(world):
    """Rows travel to the owner of their sorted position and back (two all-to-alls), nothing of grid
    size is replicated: safe set, refinement array and c_max equal the oracle's loop."""
    port = _free_port()
    results = mp.Manager().dict()
    mp.spawn(_adaptive_worker, args=(world, port, results), nprocs=world, join=True)
    for rank in range(world):
        assert results[rank] == [], results[rank]


def test_adaptive_rule_on_synthetic_cells_with_passing_batches():
    """Inputs under which many batches are accepted through refinement before one ends the loop
    (the oracle's models hardly produce that): NumPy engine + adaptive_rule against the reference's
    loop restated on arrays."""
    from np_shard_engine import (NumpyAdaptiveEngine, NumpyShardEngine, reference_adaptive_loop,
                                 synthetic_adaptive_cells)
    from safe_learning_amd.lyapunov import adaptive_rule
    refined_total = passed_first = 0
    for n, batch, max_ref, hard, seed in ((5000, 64, 3, 0.002, 1), (3000, 100, 3, 0.0, 2), (700, 50, 2, 0.01, 3),
                                          (4096, 256, 5, 0.001, 4)):
        values, decrease, thr0, tau, init = synthetic_adaptive_cells(n, seed, hard=hard)
        prior_safe, prior_ref = None, None
        for shrink in (True, False):
            p_safe = init if shrink else prior_safe
            p_ref = init.astype(np.int64) if shrink else prior_ref
            want = reference_adaptive_loop(values, decrease, thr0, tau, init, p_safe, p_ref, batch, max_ref, 1.1)
            shard = NumpyShardEngine(0, n, values, np.zeros(n, dtype=bool), init, init)
            engine = NumpyAdaptiveEngine(0, n, values, decrease, thr0, tau, init,
                                         None if shrink else prior_safe, None if shrink else prior_ref, shard)
            stats = {}
            c_max = adaptive_rule(engine, n, batch, max_ref, 1.1, stats)
            assert np.array_equal(engine.safe, want[0]) and np.array_equal(engine.refinement, want[1])
            assert c_max == want[2] or (np.isnan(c_max) and np.isnan(want[2]))
            if hard > 0:
                assert stats["b_star"] < -(-n // batch)                 # the loop ended early ...
                passed_first = max(passed_first, stats["b_star"])
            assert (want[1] > 1).sum() >= 3
            refined_total += int((want[1] > 1).sum())
            prior_safe, prior_ref = want[0].copy(), want[1].copy()
            prior_safe[np.random.default_rng(seed).choice(n, 30)] = True       # hand-marked cells
    assert refined_total > 2000                     # the all-passing case refines half of its cells
    assert passed_first >= 1                         # ... after whole batches passed through refinement
