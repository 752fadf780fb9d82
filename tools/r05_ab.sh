#!/bin/bash
# Round 5: A/B of the library in the tree against safe_learning_amd/libslhip_prev.so
# (tools/build_variant.sh): k_gp_sweep4 parity tests on the tree, alternating runs at 64^4, both
# libraries at 128^4.   tools/r05_ab.sh <label> [pytest|nopytest] [extra env for a third leg]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_ab_$1; mkdir -p $O
if [ "${2:-pytest}" = pytest ]; then
  timeout 900 python -m pytest tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py tests/test_gpu_configs.py -q -x > $O/pytest.log 2>&1
  tail -3 $O/pytest.log
fi
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f kernel_ms %.2f frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))
if not ok: print('  failed')
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
for rep in 1 2 3; do
  echo "tree"; timeout 200 $B 2>/dev/null | line
  echo "prev"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_prev.so timeout 200 $B 2>/dev/null | line
  if [ -n "$3" ]; then echo "tree, $3"; env $3 timeout 200 $B 2>/dev/null | line; fi
done
echo "tree, 128^4"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line
echo "prev, 128^4"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_prev.so timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line
} | tee $O/ab.txt
