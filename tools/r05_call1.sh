#!/bin/bash
# Round 5, GPU call 1: which memory-side counters does this rocprofv3 offer (HBM vs Infinity Cache),
# then parity + A/B of the diagonal-block split of k_gp_sweep4.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
(cd /tmp && timeout 120 rocprofv3 --list-avail > $OLDPWD/gpurun_out/r05/list_avail.txt 2>&1)
grep -i -E "mall|dram|hbm|umc|EA0?_RDREQ|EA0?_WRREQ|_EA_|MC_RD|MC_WR" gpurun_out/r05/list_avail.txt | cut -c1-200 | sort -u | head -80 > gpurun_out/r05/memory_counters.txt
wc -l gpurun_out/r05/list_avail.txt gpurun_out/r05/memory_counters.txt
tools/r05_ab.sh diag
