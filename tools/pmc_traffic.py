"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py` into
profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so it is doubled."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(db, counter, kernel_substr):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? "
        "group by kernel_name", (counter,)))
    for name, total, count in rows:
        if kernel_substr in name:
            return name, total / count
    raise SystemExit("kernel %s not found in %s" % (kernel_substr, db))


def kernel_source_sha():
    """sha256 of the kernel's source: bench.py and tests/test_bench_host.py refuse a measurement
    that was taken on another version of sl_gp4.hip."""
    path = os.path.join(ROOT, "safe_learning_amd", "csrc", "sl_gp4.hip")
    with open(path, "rb") as handle:
        return hashlib.sha256(handle.read()).hexdigest()


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "cartpole 128^4, 1024-point GP"
    kname, fetch_kib = per_dispatch(fetch_db, "FETCH_SIZE", "k_gp_sweep")
    _, write_kib = per_dispatch(write_db, "WRITE_SIZE", "k_gp_sweep")
    out = {"workload": workload, "kernel": kname.split("(")[0],
           "fetch_size_kib_reported": fetch_kib, "write_size_kib_reported": write_kib,
           "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
           "source_sha256": kernel_source_sha(),
           "note": "L2<->fabric bytes per launch of the GP sweep kernel: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                   "(KiB units, FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes): L2 misses "
                   "of the inverse Cholesky factor's fragments served by the Infinity Cache (an upper bound "
                   "of the HBM reads) + mask words and set-up scratch; profiles/README.md."}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()
