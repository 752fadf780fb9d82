#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
B="python bench.py --num-points 64 --steps 1 --warmup 0 --no-cpu-baseline"
: > gpurun_out/r03_d64.txt
for skip in 16 0; do
    echo "== tm skip $skip" >> gpurun_out/r03_d64.txt
    SL_GP4_SKIP=$skip SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_tm.so $B 2>&1 | grep "GP4TIMING\|^{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['safe_cells'])
    else: print(l.rstrip())" >> gpurun_out/r03_d64.txt
done
cat gpurun_out/r03_d64.txt
