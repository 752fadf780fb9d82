#!/bin/bash
# Round 6, GPU call 17: instruction counts and wait cycles of k_gp_small with phases switched off
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
export SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_gpsdiag.so
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt; wc -l $OUT/sq_counters.txt
one() { timeout 300 python bench.py --config $1 --steps 6 --warmup 2 --no-cpu-baseline --diagnostic 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['config']['name'], round(r['kernel_ms'],3))"; }
for f in 0 15 31; do SL_GPS_FLAGS=$f one C2-table-large "flags=$f"; done
B="python bench.py --config C2-table-large --steps 4 --warmup 1 --no-cpu-baseline --diagnostic"
for f in 0 31; do
export SL_GPS_FLAGS=$f
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR -d $OUT/ga_$f -o p -- $B > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/gb_$f -o p -- $B > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $OUT/gc_$f -o p -- $B > /dev/null 2>&1
echo "== flags $f"; python tools/pmc_dump.py k_gp_small $(find $OUT/ga_$f $OUT/gb_$f $OUT/gc_$f -name "*_results.db")
rm -rf $OUT/ga_$f $OUT/gb_$f $OUT/gc_$f
done
