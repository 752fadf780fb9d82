#!/bin/bash
# Round 5, GPU call 12: what the L2 can tell about k_gp_sweep4's traffic (64^4) - hit rate, request
# sizes, requests addressed to local memory.  Two or three counters per pass, every pass under
# timeout (a pass with five TCC_EA0 counters aborted and hung in call 6).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r05_prof; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --num-points 64"
k=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"; do
  k=$((k+1))
  timeout -k 5 100 rocprofv3 --pmc $set -d $OUT/l2_$k -o p -- $B > $OUT/l2_$k.log 2>&1
  echo "pass $k ($set): rc=$?"
done
python tools/pmc_dump.py k_gp_sweep $(find $OUT/l2_* -name "*_results.db") > $OUT/r05_pmc_l2_64.txt 2>&1
cat $OUT/r05_pmc_l2_64.txt
rm -rf $OUT/l2_?
