#!/bin/bash
# Counters of the policy-evaluation kernels (tools/gpu_configs_probe.py, C5 section)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_policy_pmc
rm -rf $OUT; mkdir -p $OUT
B="python tools/gpu_configs_probe.py"
export SL_CONFIGS=C5 SL_C5_SHORT=1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
    -d $OUT/pmc_a -o p -- $B > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_b -o p -- $B > $OUT/pmc_b.log 2>&1
python tools/pmc_dump.py k_bellman4 $(find $OUT/pmc_a $OUT/pmc_b -name "*_results.db") > $OUT/policy_pmc.txt 2>&1
rm -rf $OUT/pmc_a $OUT/pmc_b
