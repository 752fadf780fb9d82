#!/bin/bash
# Round 6, GPU call 14: what bounds the bounded streaming pass (k_finalize_dev<4> at 128^4, C4-lin)?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
B="python bench.py --config C4-lin --steps 6 --warmup 2 --no-cpu-baseline"
timeout -k 5 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/fin_a -o p -- $B > $OUT/fin_a.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 -d $OUT/fin_b -o p -- $B > $OUT/fin_b.log 2>&1
python tools/pmc_dump.py k_finalize $(find $OUT/fin_a $OUT/fin_b -name "*_results.db")
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/fin_t -o t -- $B > $OUT/fin_t.log 2>&1
python tools/kernel_stats.py $(find $OUT/fin_t -name "*_results.db" | head -1) | head -8
rm -rf $OUT/fin_a $OUT/fin_b $OUT/fin_t
