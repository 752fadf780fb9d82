#!/bin/bash
# Per-kernel times and matrix-pipe counters of the C5 bench line, final kernels of the round
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_c5_final
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --max-sweeps 8"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -8 > $OUT/C5_kernel_stats.md
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
    -d $OUT/pmc_a -o p -- $B > $OUT/pmc_a.log 2>&1
python tools/pmc_dump.py k_ $(find $OUT/pmc_a -name "*_results.db") > $OUT/C5_pmc.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_a
grep -h '^{' $OUT/trace.log | cut -c1-2500 > $OUT/C5_bench.jsonl
