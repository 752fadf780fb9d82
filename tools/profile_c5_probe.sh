#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_c5   # per-kernel times of the C5 probe (max sweep, policy evaluation sweeps)
rm -rf $OUT; mkdir -p $OUT
SL_CONFIGS=C5 SL_C5_SHORT=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python tools/gpu_configs_probe.py > $OUT/run.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -16 > gpurun_out/prof_c5_stats.md
rm -rf $OUT/trace
