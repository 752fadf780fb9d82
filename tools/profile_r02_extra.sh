#!/bin/bash
# Round-2 profiles of the other shipped kernels (k_bellman_mfma at C5, k_det_sweep at C4-lin and
# C4-det, the table flavour of k_gp_sweep at C2-table-large): rocprofv3 kernel statistics and two
# SQ counter passes each (counters only, separate runs).  Output: gpurun_out/r02_extra/*.md|txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r02_extra
rm -rf $OUT; mkdir -p $OUT
PMC_A="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
PMC_B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU"
for spec in "C5:k_bellman_mfma:--max-sweeps 6" "C4-lin:k_det_sweep:" "C4-det:k_det_sweep:" "C2-table-large:k_gp_sweep:"; do
  cfg=${spec%%:*}; rest=${spec#*:}; kern=${rest%%:*}; extra=${rest#*:}
  B="python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline $extra"
  rocprofv3 --kernel-trace --stats -d $OUT/${cfg}_trace -o t -- $B > $OUT/${cfg}_trace.log 2>&1
  rocprofv3 --pmc $PMC_A -d $OUT/${cfg}_pmc_a -o p -- $B > $OUT/${cfg}_pmc_a.log 2>&1
  rocprofv3 --pmc $PMC_B -d $OUT/${cfg}_pmc_b -o p -- $B > $OUT/${cfg}_pmc_b.log 2>&1
  T=$(find $OUT/${cfg}_trace -name "*_results.db" | head -1)
  python tools/kernel_stats.py $T > $OUT/${cfg}_kernel_stats.md 2>&1
  python tools/pmc_dump.py $kern $(find $OUT/${cfg}_pmc_a $OUT/${cfg}_pmc_b -name "*_results.db") > $OUT/${cfg}_pmc.txt 2>&1
  find $OUT/${cfg}_trace $OUT/${cfg}_pmc_a $OUT/${cfg}_pmc_b -name "*.db" -delete
  echo "== $cfg"; head -6 $OUT/${cfg}_kernel_stats.md; cat $OUT/${cfg}_pmc.txt
done
