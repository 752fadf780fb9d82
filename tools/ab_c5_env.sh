#!/bin/bash
# A/B of one environment variable on the C5 bench line, alternating inside one call:
#   tools/ab_c5_env.sh VAR v1 v2 ...   ->  gpurun_out/ab_c5_env.txt
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; shift
OUT=gpurun_out/ab_c5_env.txt
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 14 2>/dev/null | grep '^{' | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'])" >> $OUT
  done
done
cat $OUT
