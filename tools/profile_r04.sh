#!/bin/bash
# Round-4 profiles of the shipped kernels (run on the GPU box through gpurun; the summaries land in
# gpurun_out/r04_prof and are copied into profiles/ by hand).  Counters in their own passes, never
# together with a trace (gpurun refuses the combination).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04_prof
mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
# 1. kernel trace + stats of the default bench command (C4) and of C4-lin / C5
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) > $OUT/r04_kernel_stats.md 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_lin -o t -- python bench.py --config C4-lin --steps 8 --warmup 2 --no-cpu-baseline > $OUT/trace_lin.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace_lin -name "*_results.db" | head -1) > $OUT/r04_C4-lin_kernel_stats.md 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_c5 -o t -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 12 > $OUT/trace_c5.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace_c5 -name "*_results.db" | head -1) > $OUT/r04_C5_kernel_stats.md 2>&1
# 2. fabric traffic of the headline launch (128^4): FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*_results.db" | head -1) $(find $OUT/pmc_write -name "*_results.db" | head -1) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/r04_pmc_128.txt 2>&1
# 3. matrix-pipe counters at 48^4
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    -d $OUT/pmc_a -o p -- $B --num-points 48 > $OUT/pmc_a.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_a -name "*_results.db") > $OUT/r04_pmc_48.txt 2>&1
# 4. the VALU-bound step: issue utilisation of k_det_rows / k_finalize_dev (C4-lin)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY \
    -d $OUT/pmc_lin -o p -- python bench.py --config C4-lin --steps 8 --warmup 2 --no-cpu-baseline > $OUT/pmc_lin.log 2>&1
python tools/pmc_dump.py k_ $(find $OUT/pmc_lin -name "*_results.db") > $OUT/r04_C4-lin_pmc.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_lin_f -o f -- python bench.py --config C4-lin --steps 8 --warmup 2 --no-cpu-baseline > $OUT/pmc_lin_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_lin_w -o w -- python bench.py --config C4-lin --steps 8 --warmup 2 --no-cpu-baseline > $OUT/pmc_lin_w.log 2>&1
python tools/pmc_dump.py k_ $(find $OUT/pmc_lin_f $OUT/pmc_lin_w -name "*_results.db") >> $OUT/r04_C4-lin_pmc.txt 2>&1
# 5. C5: matrix pipe
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 \
    -d $OUT/pmc_c5 -o p -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 12 > $OUT/pmc_c5.log 2>&1
python tools/pmc_dump.py k_bellman $(find $OUT/pmc_c5 -name "*_results.db") > $OUT/r04_C5_pmc.txt 2>&1
grep -h '^{' $OUT/trace.log $OUT/trace_lin.log $OUT/trace_c5.log | cut -c1-600 > $OUT/r04_bench_lines.txt
rm -rf $OUT/trace $OUT/trace_lin $OUT/trace_c5 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_a $OUT/pmc_lin $OUT/pmc_lin_f $OUT/pmc_lin_w $OUT/pmc_c5
cat $OUT/r04_kernel_stats.md $OUT/r04_C4-lin_kernel_stats.md $OUT/r04_C5_kernel_stats.md $OUT/r04_pmc_128.txt $OUT/r04_pmc_48.txt $OUT/r04_C4-lin_pmc.txt $OUT/r04_C5_pmc.txt
