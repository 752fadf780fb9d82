#!/bin/bash
# Round 6, GPU call 19: the floor of k_gp_small - what a sweep costs with the GEMM passes, the check and
# the policy switched off (development build), against the grid size.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_gpsdiag.so
one() { timeout 300 python bench.py --config $1 --steps 6 --warmup 2 --no-cpu-baseline --diagnostic 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['config']['name'], round(r['kernel_ms'],3))"; }
for f in 0 15 31 32 40 56 63; do SL_GPS_FLAGS=$f one C2-table-large "flags=$f"; done
for f in 0 32 56; do SL_GPS_FLAGS=$f SL_GP_SMALL_WAVES=8 one C2-table-stack "stack flags=$f"; done
