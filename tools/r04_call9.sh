#!/bin/bash
# Round 4, call 9: where the generation phase of k_gp_sweep4 spends its time (diagnostic builds,
# results meaningless): 1 = no loads / exponentials, 2 = no recurrence, 4 = one store instead of 16.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call9; mkdir -p $O
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f  kernel_ms %.2f  frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
echo "shipped"; timeout 200 $B 2>/dev/null | line
for v in 1 2 4 7; do
  echo "diag $v"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_diag$v.so timeout 200 $B 2>/dev/null | line
done
echo "shipped, skip 1"; SL_GP4_SKIP=1 timeout 200 $B 2>/dev/null | line
for skip in 0 1 3 8 15; do
  echo "one workgroup per CU, skip $skip"; SL_GP4_WGS=1 SL_GP4_SKIP=$skip timeout 200 $B 2>/dev/null | line
done
echo "four workgroups per CU requested (two resident), skip 0"; SL_GP4_WGS=4 timeout 200 $B 2>/dev/null | line
} | tee $O/gen_diag.txt
