#!/bin/bash
# Per-kernel times and counters of the C5 bench line (Bellman max sweep), shipped kernels
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_c5_bench
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --max-sweeps 8"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -8 > $OUT/C5_kernel_stats.md
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
    -d $OUT/pmc_a -o p -- $B > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_b -o p -- $B > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/pmc_c -o p -- $B > $OUT/pmc_c.log 2>&1
python tools/pmc_dump.py k_ $(find $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c -name "*_results.db") > $OUT/C5_pmc.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
grep -h '^{' $OUT/trace.log | cut -c1-2500 > $OUT/C5_bench.jsonl
python bench.py --config C5 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-2500 > $OUT/C5_bench_full.jsonl
