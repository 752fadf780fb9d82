#!/bin/bash
# Kernel trace of C3 (pendulum 2048^2, 2048-point GP, LyapunovNetwork): what the second pass costs.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_c3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --config C3 --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
python tools/kernel_stats.py $(find $O/trace -name "*_results.db" | head -1) > $O/r04_C3_kernel_stats.md 2>&1
rm -rf $O/trace
head -8 $O/r04_C3_kernel_stats.md | cut -c1-200
