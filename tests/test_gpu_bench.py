"""bench.py on the GPU box: the single-rank line and the self-launched two-rank run (both ranks
on cuda:0, gloo rendezvous on 127.0.0.1) must describe the same result."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*flags, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_line_small_and_two_ranks():
    small = ["--num-points", "24", "--n-gp", "300", "--steps", "2", "--warmup", "1"]
    one = _run(*small)
    assert one["n_gpus"] == 1 and one["value"] > 0
    for key in ("metric", "unit", "ms_per_step", "roofline", "cpu_baseline", "end_to_end_ms"):
        assert key in one
    assert one["roofline"]["bound"] == "mfma" and 0 < one["roofline"]["frac"] < 1
    cpu = one["cpu_baseline"]
    for key in ("value", "cores", "kind", "sample", "threads4_value", "reference_faithful_ms",
                "cpu_model", "blas"):
        assert key in cpu
    cfg = one["config"]
    # the workload is not degenerate: cells pass the check and the level set grows
    assert 0 < cfg["negative_cells"] < 24 ** 4
    assert cfg["safe_cells"] >= cfg["initial_cells"] + 50
    two = _run("--gpus", "2", "--no-cpu-baseline", *small)
    assert two["n_gpus"] == 2 and two["config"]["collectives"]["world_size"] == 2
    assert len(two["config"]["per_rank_kernel_ms"]) == 2
    assert two["config"]["safe_cells"] == cfg["safe_cells"]
    assert two["config"]["negative_cells"] == cfg["negative_cells"]
    assert two["config"]["c_max"] == cfg["c_max"]


def test_bench_rccl_call_sequence_at_world_one():
    """The exact collective sequence of the N > 1 path on this one-GPU box: SL_FORCE_COLLECTIVES=1
    makes bench.py create an `nccl` (= RCCL) process group of one rank and the package issue
    every gather / reduction of the sharded path as an RCCL call on device tensors
    (two packed-record gathers + the mask-word gather per update, single-tensor gathers).  The
    result must equal the plain single-rank run and the line must carry the collective timing."""
    small = ["--num-points", "24", "--n-gp", "300", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    plain = _run(*small)
    forced = _run("--backend", "nccl", *small, SL_FORCE_COLLECTIVES="1")
    cfg, ref = forced["config"], plain["config"]
    assert cfg["collectives"] == {"backend": "nccl", "world_size": 1, "ranks_share_a_device": False}
    assert ref["collectives"]["backend"] is None and ref["collective_ms"] == 0.0
    assert cfg["collective_ms"] > 0.0 and len(cfg["per_rank_collective_ms"]) == 1
    assert len(cfg["per_rank_ms_per_step"]) == 1 and cfg["per_rank_ms_per_step"][0] > 0
    for key in ("safe_cells", "negative_cells", "c_max", "initial_cells"):
        assert cfg[key] == ref[key]
    assert forced["roofline"]["kernel"].startswith("k_gp_sweep4<")
    # the Bellman path: table gather + residual reduction over RCCL
    c5 = ["--config", "C5", "--num-points", "10", "--n-gp", "128", "--steps", "3", "--no-cpu-baseline",
          "--max-sweeps", "40"]
    plain5 = _run(*c5)
    forced5 = _run("--backend", "nccl", *c5, SL_FORCE_COLLECTIVES="1")
    assert forced5["config"]["relative_residual"] == plain5["config"]["relative_residual"]
    assert forced5["config"]["collective_ms"] > 0.0


def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1 (one process per rank started by
    torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE in the environment); on this one-GPU box
    both ranks share cuda:0 and rendezvous over gloo."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--num-points", "24", "--n-gp", "300", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["collectives"]["world_size"] == 2
    assert out["config"]["safe_cells"] > out["config"]["initial_cells"]


@pytest.mark.parametrize("config,flags", [
    ("C1", []),
    ("C2", ["--num-points", "64", "--n-gp", "128"]),
    ("C4-lin", ["--num-points", "24"]),
    ("C4-det", ["--num-points", "16"]),
])
def test_bench_other_lyapunov_configs(config, flags):
    out = _run("--config", config, "--no-cpu-baseline", *flags)
    assert out["config"]["name"] == config and out["value"] > 0
    assert out["roofline"]["bound"] == ("mfma" if config == "C2" else "hbm")


def test_bench_c5_policy_line():
    """--config C5-policy: the policy-evaluation sweep with the greedy table policy (on a 16-cell
    last axis this is k_bellman_policy_mfma; the 64-cell shapes are tests/test_gpu_rl.py's)."""
    out = _run("--config", "C5-policy", "--num-points", "10", "--n-gp", "128", "--steps", "3")
    assert out["config"]["name"] == "C5-policy" and out["unit"] == "vertices/s" and out["value"] > 0
    assert out["config"]["distinct_actions_per_row"] >= 1.0
    assert "k_bellman" in out["roofline"]["kernel"] and "cpu_baseline" not in out


def test_bench_c5_runs_to_convergence():
    out = _run("--config", "C5", "--num-points", "10", "--n-gp", "128", "--steps", "3",
               "--no-cpu-baseline", "--max-sweeps", "2500")
    cfg = out["config"]
    assert cfg["converged"] and cfg["relative_residual"] <= 1e-6
    assert cfg["residual_monotone"]
    assert 100 < cfg["sweeps_to_convergence"] < 2500      # gamma = 0.98
    two = _run("--config", "C5", "--num-points", "10", "--n-gp", "128", "--steps", "3",
               "--no-cpu-baseline", "--max-sweeps", "2500", "--gpus", "2")
    assert two["config"]["sweeps_to_convergence"] == cfg["sweeps_to_convergence"]
