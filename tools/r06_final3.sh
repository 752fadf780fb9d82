#!/bin/bash
# Round 6: the driver's bench command and the rocprofv3 kernel trace of the same kernel on ONE box
# (the boxes of the pool differ by 1 - 4 %: a trace from one box does not match a bench line from another).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_final3; mkdir -p $OUT
(time timeout -k 5 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/r06_bench_driver_command.log 2>&1
echo "driver bench rc=$?"
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) > $OUT/r06_kernel_stats.md 2>&1
rm -rf $OUT/trace
head -5 $OUT/r06_kernel_stats.md
grep -o '"kernel_ms": [0-9.]*' $OUT/r06_bench_driver_command.log
