#!/bin/bash
# Round 6, GPU call 13: the streaming pass skips rows provably above the level (SlRowValues::eight_bounded):
# whole suite, then the step times of the configurations whose step it is a visible part of.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/call13_all.log 2>&1
echo "all rc=$?"; tail -4 $OUT/call13_all.log
bash tools/bench_configs.sh C1 C2 C4-lin C4-det 2>&1 | tail -4
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["config"]["name"], "step", round(d["ms_per_step"],4), "sweep", round(r["kernel_ms"],4), "finalize", round(r.get("finalize_ms",0),4), "step_frac", r.get("step_frac"))
PY
