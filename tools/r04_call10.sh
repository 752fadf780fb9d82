#!/bin/bash
# Round 4, call 10: A/B of a k_gp_sweep4 variant library (safe_learning_amd/libslhip_$1.so)
# against the library in the tree: parity tests on the variant, then alternating runs at 64^4.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
V=${1:-early}
O=gpurun_out/r04_ab_$V; mkdir -p $O
VL=$PWD/safe_learning_amd/libslhip_$V.so
SL_LIB_PATH=$VL timeout 900 python -m pytest tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f  kernel_ms %.2f  frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
for rep in 1 2 3; do
  echo "tree"; timeout 200 $B 2>/dev/null | line
  echo "$V"; SL_LIB_PATH=$VL timeout 200 $B 2>/dev/null | line
done
for skip in 1 2 3; do echo "$V, skip $skip"; SL_LIB_PATH=$VL SL_GP4_SKIP=$skip timeout 200 $B 2>/dev/null | line; done
echo "$V, one workgroup per CU"; SL_LIB_PATH=$VL SL_GP4_WGS=1 timeout 200 $B 2>/dev/null | line
} | tee $O/ab.txt
