#!/bin/bash
# Round 4, call 13: the second workgroup of a CU generates the next chunk BEFORE its MFMA stream,
# the first one after it (SL_GP4_PRIO=0: both after).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py tests/test_gpu_configs.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
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
  echo "roles"; timeout 200 $B 2>/dev/null | line
  echo "no roles"; SL_GP4_PRIO=0 timeout 200 $B 2>/dev/null | line
done
for skip in 1 3; do echo "roles, skip $skip"; SL_GP4_SKIP=$skip timeout 200 $B 2>/dev/null | line; done
} | tee $O/ab.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line | tee -a $O/ab.txt
