#!/bin/bash
# Round-3 profiles of the other shipped kernels: k_gp_small (C2-table-large), k_det_rows (C4-lin),
# the Bellman sweep (C5) and the survey-hyper-parameter variant of the headline (cost neutrality).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03_extra
mkdir -p $OUT
for cfg in C2-table-large C4-lin C5; do
  B="python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --max-sweeps 8"
  rocprofv3 --kernel-trace --stats -d $OUT/${cfg}_trace -o t -- $B > $OUT/${cfg}_trace.log 2>&1
  python tools/kernel_stats.py $(find $OUT/${cfg}_trace -name "*_results.db" | head -1) | head -6 > $OUT/${cfg}_kernel_stats.md
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
      -d $OUT/${cfg}_pmc_a -o p -- $B > $OUT/${cfg}_pmc_a.log 2>&1
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/${cfg}_pmc_b -o p -- $B > $OUT/${cfg}_pmc_b.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/${cfg}_pmc_c -o p -- $B > $OUT/${cfg}_pmc_c.log 2>&1
  python tools/pmc_dump.py k_ $(find $OUT/${cfg}_pmc_a $OUT/${cfg}_pmc_b $OUT/${cfg}_pmc_c -name "*_results.db") > $OUT/${cfg}_pmc.txt 2>&1
  rm -rf $OUT/${cfg}_trace $OUT/${cfg}_pmc_a $OUT/${cfg}_pmc_b $OUT/${cfg}_pmc_c
done
# SURVEY 8d's literal hyper-parameters (one cell of 2.7e8 passes): same kernel, same cost
python bench.py --gp-variant survey --steps 3 --warmup 1 --no-cpu-baseline > $OUT/survey_variant.log 2>&1
grep -h '^{' $OUT/survey_variant.log | cut -c1-1500
cat $OUT/*_kernel_stats.md
