#!/bin/bash
# Where the time of k_gp_small goes: a development build (tools/build_variant.sh gpsdiag sl_gp_small
# sl_gp_small.hip, run here first) with phases switched off - SL_GPS_FLAGS: 1 no kernel evaluation,
# 2 no MFMAs, 4 no fragment loads, 8 no check, 16 no policy, 32 no GEMM passes.  The results of such
# runs are meaningless, the durations are the point (profiles/dropped/r06_gp_small_loop_variants.txt).
# Run on the GPU box:  gpurun -- 'bash tools/gp_small_attribution.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_gpsdiag.so
one() { timeout 300 python bench.py --config $1 --steps 6 --warmup 2 --no-cpu-baseline --diagnostic 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['config']['name'], round(r['kernel_ms'],3), r.get('kernel'))"; }
for c in C2-table-large C2-notebook; do
  for split in 0 1; do
    for f in 0 1 2 4 6 7 8 15 31 32 63; do
      SL_GP_SMALL_SPLIT=$split SL_GPS_FLAGS=$f one $c "split=$split flags=$f"
    done
  done
done
