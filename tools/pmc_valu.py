"""Vector-ALU issue utilisation of the kernels that no byte or matrix-pipe roof describes (the
deterministic-dynamics sweeps, k_gp_small) -> profiles/pmc_valu.json, which bench.py reports as
`roofline` with `bound: "valu"`.

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
        SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES -d <dir> -o p -- python bench.py --config <cfg> ...
    rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
        SQ_INSTS_VALU -d <dir2> -o q -- python bench.py --config <cfg> ...
    python tools/pmc_valu.py <cfg> <kernel substring> <p_results.db> [<q_results.db>]

issue utilisation = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024): SQ_ACTIVE_INST_VALU counts
(in quad-cycles, summed over the XCDs' SQs) the cycles a SIMD is issuing a vector-ALU instruction;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs (256 CUs x 4).  The entry is tied to the
kernel sources by a sha256 like profiles/pmc_traffic.json is; bench.py refuses a stale one."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# sources whose change invalidates an entry, per kernel family
SOURCES = {"k_det": ["sl_det_rows.hip", "sl_kernels.hip", "sl_model.h", "sl_common.h"],
           "k_bellman": ["sl_bellman4.hip", "sl_bellman.hip", "sl_model.h", "sl_common.h"],
           "k_gp_small": ["sl_gp_small.hip", "sl_model.h", "sl_common.h"],
           "k_finalize": ["sl_level.hip", "sl_model.h", "sl_common.h"]}


def sources_sha(kernel):
    family = next((k for k in SOURCES if kernel.startswith(k) or k in kernel), None)
    if family is None:
        return None
    h = hashlib.sha256()
    for name in SOURCES[family]:
        with open(os.path.join(ROOT, "safe_learning_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def counters(db, substr):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                       "from counters_collection group by kernel_name, counter_name")
    best = {}
    for name, counter, total, ndisp in rows:
        if substr in name:
            short = name.replace("(anonymous namespace)::", "").split("(")[0]
            best.setdefault(short, {})[counter] = total / max(ndisp, 1)
    if not best:
        raise SystemExit("no kernel with %r in %s" % (substr, db))
    # the dominant kernel of the family: the one with the most busy cycles
    name = max(best, key=lambda k: best[k].get("GRBM_GUI_ACTIVE", best[k].get("SQ_INSTS_VALU", 0.0)))
    return name, best[name]


def main():
    cfg, substr, db = sys.argv[1:4]
    name, c = counters(db, substr)
    entry = {"config": cfg, "kernel": name.replace("void ", ""), "counters": c}
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    entry["valu_issue_utilisation"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        entry["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        entry["waves_waiting"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    if len(sys.argv) > 4:
        _, q = counters(sys.argv[4], name.split("<")[0].replace("void ", "").strip())
        entry["counters_fp64"] = q
        fp64 = sum(q.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                            "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        if q.get("SQ_INSTS_VALU"):
            entry["fp64_share_of_valu_instructions"] = fp64 / q["SQ_INSTS_VALU"]
    entry["source_sha256"] = sources_sha(entry["kernel"])
    path = os.path.join(ROOT, "profiles", "pmc_valu.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {}
    table[cfg] = entry
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
