#!/bin/bash
# One bench.py line per BASELINE configuration -> gpurun_out/configs.jsonl (copied to profiles/rNN_configs.jsonl) (+ a short table).
mkdir -p gpurun_out
: > gpurun_out/configs.jsonl
for c in ${@:-C1 C2 C2-table C2-table-large C2-table-stack C2-notebook C2-table-det C3 C4-lin C4-det C5 C5-policy}; do
  # steps: 3 (C3, 0.27 s each), 10 (10 ms and more), 200 for the sub-10-ms configurations - 10 steps of a 0.4 ms
  # update end before the clocks have settled (C2: 0.37 ms kernel over 10 steps, 0.33 over 200)
  steps=200; warm=20
  case $c in C3) steps=3; warm=2;; C4-det|C5) steps=10; warm=2;; esac
  timeout 600 python bench.py --config $c --steps $steps --warmup $warm --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/configs.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); r = d["roofline"]; c = d["config"]
    print("%-14s ms/step %10.3f  kernel_ms %10.3f  %s frac %.3f  value %.4g %s  %s" % (
        c["name"], d["ms_per_step"], r["kernel_ms"], r["bound"], r["frac"], d["value"], d["unit"],
        {k: c[k] for k in ("safe_cells", "initial_cells", "sweeps_to_convergence", "converged") if k in c}))
PY
