#!/bin/bash
# Round 5, GPU call 10: SURVEY 8d's 2^24-cell CPU-baseline sample beside the headline line, and a
# two-rank line on the one GPU (gloo) that shows the per-rank fields bench.py checks since round 5.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 300 python bench.py --config C2 --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/r05_bench_two_ranks_one_gpu.log 2>$O/two.err
cut -c1-1400 $O/r05_bench_two_ranks_one_gpu.log
timeout 700 python bench.py --steps 2 --warmup 1 --cpu-cells 16777216 > $O/r05_cpu_baseline.log 2>$O/cpu.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_final/r05_cpu_baseline.log") if l.startswith("{")][0])
print(json.dumps(d["cpu_baseline"], indent=1))
PY
