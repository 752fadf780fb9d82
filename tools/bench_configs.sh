#!/bin/bash
# One bench.py line per BASELINE configuration -> gpurun_out/configs.jsonl (copied to profiles/rNN_configs.jsonl) (+ a short table).
mkdir -p gpurun_out
: > gpurun_out/configs.jsonl
for c in ${@:-C1 C2 C2-table C2-table-large C2-table-stack C2-notebook C2-table-det C3 C4-lin C4-det C5 C5-policy}; do
  steps=10; [ "$c" = C3 ] && steps=3
  timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/configs.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); r = d["roofline"]; c = d["config"]
    print("%-14s ms/step %10.3f  kernel_ms %10.3f  %s frac %.3f  value %.4g %s  %s" % (
        c["name"], d["ms_per_step"], r["kernel_ms"], r["bound"], r["frac"], d["value"], d["unit"],
        {k: c[k] for k in ("safe_cells", "initial_cells", "sweeps_to_convergence", "converged") if k in c}))
PY
