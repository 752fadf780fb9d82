#!/bin/bash
# Round-3 profiles of the shipped kernels (run on the GPU box through gpurun; results under
# gpurun_out/r03_prof, summarised into profiles/r03_summary.md by hand + tools/*.py).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03_prof
mkdir -p $OUT
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
# 1. kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
# 2. SQ / TCC counter passes at 48^4 (separate runs, counters only)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    -d $OUT/pmc_a -o p -- $B --num-points 48 > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
    -d $OUT/pmc_b -o p -- $B --num-points 48 > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_c -o p -- $B --num-points 48 > $OUT/pmc_c.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/pmc_d -o p -- $B --num-points 48 > $OUT/pmc_d.log 2>&1
# 3. HBM-side traffic of the headline launch (128^4)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $B > $OUT/pmc_write.log 2>&1
# summaries
find $OUT -name "*_results.db" | sort > $OUT/dbs.txt
T=$(find $OUT/trace -name "*_results.db" | head -1)
python tools/kernel_stats.py $T > $OUT/kernel_stats.md 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d -name "*_results.db") > $OUT/pmc_48.txt 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/pmc_128.txt 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*_results.db" | head -1) $(find $OUT/pmc_write -name "*_results.db" | head -1) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
grep -h '^{' $OUT/trace.log $OUT/pmc_a.log | cut -c1-400 > $OUT/bench_lines.txt
# the result databases are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf $OUT/trace $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/kernel_stats.md $OUT/pmc_48.txt $OUT/pmc_128.txt
