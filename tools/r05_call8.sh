#!/bin/bash
# Round 5, GPU call 8: kernel stats + matrix-pipe counters of the C5 sweep, vector-instruction
# counters of k_gp_sweep4 at 48^4 (every rocprofv3 under timeout).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash tools/profile_r05.sh c5 > gpurun_out/r05_prof_c5.log 2>&1
tail -12 gpurun_out/r05_prof_c5.log | cut -c1-200
OUT=gpurun_out/r05_prof
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM \
    -d $OUT/pmc_b -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --num-points 48 > $OUT/pmc_b.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_b -name "*_results.db") > $OUT/r05_pmc_48_valu.txt 2>&1
cat $OUT/r05_pmc_48_valu.txt; rm -rf $OUT/pmc_b
