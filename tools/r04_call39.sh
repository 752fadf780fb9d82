#!/bin/bash
# Round 4, call 39: k_gp_small on the row blocks that hold training points (tree) against all row
# blocks of the padded upload (libslhip_prev.so).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_gp_kernels.py tests/test_gpu_reference_gp.py tests/test_gpu_notebook_loop.py tests/test_gpu_lyapunov.py -q -x 2>&1 | tail -2
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('    kernel_ms %.3f  %s' % (r['kernel_ms'], r['kernel'][:70]))
if not ok: print('    failed')
"; }
P=$PWD/safe_learning_amd/libslhip_prev.so
{
for cfgargs in "C2 --num-points 1024 --n-gp 130" "C2 --num-points 1024 --n-gp 160" "C2 --num-points 1024 --n-gp 200" "C2 --num-points 1024 --n-gp 256" "C2-table-large --n-gp 130" "C2-table-stack --n-gp 100" "C2-table-stack"; do
  echo "$cfgargs"
  timeout 300 python bench.py --config $cfgargs --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
  SL_LIB_PATH=$P timeout 300 python bench.py --config $cfgargs --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
done
} | tee gpurun_out/r04_gp_small_rows.txt
