#!/bin/bash
# Round-6 profiles of the shipped kernels (run on the GPU box through gpurun; the summaries land in
# gpurun_out/r06_prof and are copied into profiles/ by hand).  Counters in their own passes, never
# together with a trace (gpurun refuses the combination).  EVERY rocprofv3 runs under `timeout`: an
# unsupported counter combination makes rocprofv3 abort inside the application and then hang in its
# own signal handler (round 5, call 6: 49 GPU-minutes lost to the TCC_EA0_RDREQ_* pass below, which is
# therefore gone - the memory-side counters that exist are listed in profiles/r06_rocprofv3_memory_counters.txt).
#   tools/profile_r06.sh [headline] [valu] [c5]        (default: all three parts)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_prof
mkdir -p $OUT
parts="${@:-headline valu c5}"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
db() { find $1 -name "*_results.db" | head -1; }
if [[ " $parts " == *" headline "* ]]; then
# 1. kernel trace + stats of the default bench command (C4)
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(db $OUT/trace) > $OUT/r06_kernel_stats.md 2>&1
# 2. fabric traffic of the headline launch (128^4): FETCH_SIZE and WRITE_SIZE in separate passes
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(db $OUT/pmc_fetch) $(db $OUT/pmc_write) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/r06_pmc_128.txt 2>&1
# 3. matrix-pipe counters at 48^4
timeout -k 5 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    -d $OUT/pmc_a -o p -- $B --num-points 48 > $OUT/pmc_a.log 2>&1
timeout -k 5 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM \
    -d $OUT/pmc_b -o p -- $B --num-points 48 > $OUT/pmc_b.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_a $OUT/pmc_b -name "*_results.db") > $OUT/r06_pmc_48.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_a $OUT/pmc_b
fi
if [[ " $parts " == *" valu "* ]]; then
# 4. the kernels whose roof is vector-ALU issue: utilisation + FP64 share -> profiles/pmc_valu.json
# (key of the entry, bench.py configuration, kernel-name substring; C5-lookup: is k_bellman_lookup bound
# by vector-ALU issue - then fusing it behind the GEMM cannot overlap the two, both use the FP64 pipe -
# or by the latency of its gathers?)
for spec in "C4-lin C4-lin k_det_rows" "C4-det C4-det k_det_sweep" "C2-table-det C2-table-det k_det" "C2-table C2-table k_gp_small" \
            "C2-table-large C2-table-large k_gp_small" "C2-table-stack C2-table-stack k_gp_small" "C2-notebook C2-notebook k_gp_small" \
            "C5-lookup C5 k_bellman_lookup"; do
  set -- $spec; key=$1; cfg=$2; sub=$3
  extra=""; [ $cfg = C5 ] && extra="--max-sweeps 12"
  timeout -k 5 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES \
      -d $OUT/v_$key -o p -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $extra > $OUT/v_$key.log 2>&1
  timeout -k 5 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
      -d $OUT/w_$key -o q -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $extra > $OUT/w_$key.log 2>&1
  python tools/pmc_valu.py $key $sub $(db $OUT/v_$key) $(db $OUT/w_$key) > $OUT/valu_$key.json 2>&1
  rm -rf $OUT/v_$key $OUT/w_$key
done
cp profiles/pmc_valu.json $OUT/pmc_valu.json
fi
if [[ " $parts " == *" c5 "* ]]; then
# 5. C5: the loop served from the successor cache (k_bellman_cached) after the recomputing first sweeps
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_c5 -o t -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 16 > $OUT/trace_c5.log 2>&1
python tools/kernel_stats.py $(db $OUT/trace_c5) > $OUT/r06_C5_kernel_stats.md 2>&1
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_c5p -o t -- python bench.py --config C5-policy --steps 8 --warmup 2 --no-cpu-baseline > $OUT/trace_c5p.log 2>&1
python tools/kernel_stats.py $(db $OUT/trace_c5p) > $OUT/r06_C5-policy_kernel_stats.md 2>&1
timeout -k 5 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    -d $OUT/pmc_c5 -o p -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 16 > $OUT/pmc_c5.log 2>&1
python tools/pmc_dump.py k_bellman $(db $OUT/pmc_c5) > $OUT/r06_C5_pmc.txt 2>&1
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_c5f -o f -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 16 > $OUT/pmc_c5f.log 2>&1
timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_c5w -o w -- python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline --max-sweeps 16 > $OUT/pmc_c5w.log 2>&1
python tools/pmc_dump.py k_bellman_cached $(db $OUT/pmc_c5f) $(db $OUT/pmc_c5w) > $OUT/r06_C5_cached_traffic.txt 2>&1
rm -rf $OUT/trace_c5 $OUT/trace_c5p $OUT/pmc_c5 $OUT/pmc_c5f $OUT/pmc_c5w
fi
