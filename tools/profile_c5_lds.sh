#!/bin/bash
# LDS counters of the C5 GEMM kernel (one pass per counter group: an unknown name only loses its pass)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_c5_lds
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --max-sweeps 5"
i=0
for set in "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p -- $B > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed.txt
done
python tools/pmc_dump.py k_bellman4s $(find $OUT -name "*_results.db") > $OUT/C5_lds_pmc.txt 2>&1
rm -rf $OUT/p?/
