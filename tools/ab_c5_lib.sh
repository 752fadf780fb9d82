#!/bin/bash
# A/B of two builds of the library on the C5 bench line, alternating inside one call:
#   tools/ab_c5_lib.sh [base.so]   ->  gpurun_out/ab_c5_lib.txt   (base default: safe_learning_amd/libslhip_base.so)
cd "${GRAFT_REPO_ROOT:-.}"
BASE=${1:-$PWD/safe_learning_amd/libslhip_base.so}
OUT=gpurun_out/ab_c5_lib.txt
: > $OUT
for rep in 1 2; do
  for lib in $BASE $PWD/safe_learning_amd/libslhip.so; do
    SL_LIB_PATH=$lib python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 14 2>/dev/null | grep '^{' | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'])" >> $OUT
  done
done
cat $OUT
