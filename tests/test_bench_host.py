"""bench.py pieces that need no GPU: argument handling, roofline arithmetic, the CPU-baseline legs
on small cases (the oracle is the thing timed there - allowed for bench.py's cpu_baseline only)."""

import os
import sys


from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_flops_per_check_contract():
    # SURVEY 8d figures: C2, C3 (without the network), C4
    assert bench.flops_per_check(512, 3, 2) == 7168 + 2048 + 263168
    assert bench.flops_per_check(1024, 5, 4) == 1081344
    assert bench.flops_per_check(2048, 3, 2) == 28672 + 8192 + 4198400
    # FunctionStack: the kernel row and the solve are repeated per head
    assert bench.flops_per_check(1024, 5, 4, heads=4) == 4 * (22528 + 1050624) + 8192


def test_args_and_workloads():
    args = bench.parse_args([])
    assert args.gpus == 1 and args.config == "C4" and args.steps >= 1
    for cfg in bench.CONFIGS:
        args = bench.parse_args(["--config", cfg, "--num-points", "6", "--n-gp", "20"])
        kind, label, case = bench.build_workload(args)
        assert kind == {"C5": "bellman", "C5-policy": "policy"}.get(cfg, "lyapunov")
        assert isinstance(label, str) and case["d"] in (1, 2, 4)
    # the headline: BASELINE.json's grid and GP sizes
    kind, label, case = bench.build_workload(bench.parse_args([]))
    assert case["num_points"] == [128] * 4 and len(case["dynamics"]["X"]) == 1024
    assert "128^4" in label and "1024-point" in label


def test_reference_faithful_leg_small():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from gp_cases import INFORMED
    case = cases.make_case("pendulum", num_points=40, n_gp=60, tau_scale=0.01, **INFORMED)
    whole = bench._reference_faithful(case)
    assert "whole grid" in whole["reference_faithful_sample"]
    assert whole["reference_faithful_safe_cells"] > 100 and whole["reference_faithful_ms"] > 0
    slab = bench._reference_faithful(case, max_cells=400)
    assert "central slab of 10 of the 40 layers" in slab["reference_faithful_sample"]
    done, seconds = bench._oracle_batches(case, 0.2, threads=2)
    assert done >= 1600 and seconds > 0
    model, blas, threads = bench._cpu_info()
    assert isinstance(model, str) and isinstance(blas, str) and threads >= 1


def test_traffic_measurement_belongs_to_the_shipped_kernel():
    """roofline.traffic is read from profiles/pmc_traffic.json (separate rocprofv3 counter passes,
    tools/profile_r05.sh): the file records the sha256 of sl_gp4.hip it was measured on, and a kernel
    edit without a new measurement fails here (bench.py then reports traffic = null)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as handle:
        pmc = json.load(handle)
    assert pmc["source_sha256"] == pmc_traffic.kernel_source_sha()
    assert pmc["bytes_per_launch"] > 2.2e9            # at least the algorithmic bytes


def test_valu_measurements_belong_to_the_shipped_kernels():
    """roofline.bound = "valu" (the deterministic sweeps, k_gp_small) reads the vector-ALU issue
    utilisation from profiles/pmc_valu.json (rocprofv3 counter passes, tools/profile_r05.sh +
    tools/pmc_valu.py).  Every entry records the sha256 of the sources of its kernel family: an edit
    without a new measurement fails here (bench.py then falls back to the byte / matrix-pipe figure
    and says why)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_valu
    with open(os.path.join(ROOT, "profiles", "pmc_valu.json")) as handle:
        table = json.load(handle)
    assert {"C4-lin", "C4-det", "C2-table-large"} <= set(table)
    for key, entry in table.items():
        assert entry["source_sha256"] == pmc_valu.sources_sha(entry["kernel"]), key
        assert 0.0 < entry["valu_issue_utilisation"] <= 1.0, key


def test_valu_roof_replaces_the_byte_roof_only_with_a_matching_measurement(tmp_path, monkeypatch):
    import json
    import bench
    args = bench.parse_args(["--config", "C4-lin"])
    base = {"bound": "hbm", "achieved": 2400.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.3, "traffic": None,
            "bytes_per_check": 10.0}
    out = bench.valu_roof(args, dict(base))
    assert out["bound"] == "valu" and out["hbm_frac"] == 0.3 and 0 < out["frac"] <= 1
    # another shape than the measured one, or a stale entry: the byte roof stays
    assert bench.valu_roof(bench.parse_args(["--config", "C4-lin", "--num-points", "64"]), dict(base))["bound"] == "hbm"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_valu
    monkeypatch.setattr(pmc_valu, "sources_sha", lambda kernel: "other")
    stale = bench.valu_roof(args, dict(base))
    assert stale["bound"] == "hbm" and "valu_note" in stale


def test_bench_refuses_diagnostic_environments():
    """A development library (SL_LIB_PATH) or the work-skipping switches of a development build
    (SL_GP4_SKIP, SL_BM_FLAGS, SL_B4P_FLAGS) do not produce benchmark lines: bench.py exits before
    it touches a GPU.  The shipped library does not read those switches at all - no `getenv` outside
    the one function that fills the context's switches at sl_ctx_create."""
    import subprocess
    for name in bench.REFUSED_ENV:
        env = dict(os.environ, **{name: "8"})
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert res.returncode != 0 and name in res.stderr and "{" not in res.stdout, (name, res.stderr[-300:])
    # every SL_* variable the sources read is either refused or listed in the line
    import glob
    import re
    csrc = os.path.join(ROOT, "safe_learning_amd", "csrc")
    calls, names = [], set()
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(path).read()
        calls += [(os.path.basename(path), m.start()) for m in re.finditer(r"\bgetenv\(", text)]
        names |= set(re.findall(r'"(SL_[A-Z0-9_]+)"', text))
    # getenv appears in sl_kernels.hip's env_int (context creation) and in the SL_DIAG-only helper
    assert sorted({f for f, _ in calls}) == ["sl_common.h", "sl_kernels.hip"], calls
    known = set(bench.REFUSED_ENV) | set(bench.AB_ENV) | {"SL_PROBE_BLOCKS_PER_CU", "SL_BELLMAN4_POLICY_VERBOSE"}
    assert names - known == set(), names - known
