#!/bin/bash
# Round 4, call 35: k_gp_small with tiles drawn from a counter (tree) against the fixed tile list
# per wavefront (libslhip_prev.so).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_gp_kernels.py tests/test_gpu_reference_gp.py tests/test_gpu_notebook_loop.py tests/test_gpu_lyapunov.py tests/test_gpu_distributed.py -q -x 2>&1 | tail -2
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
for cfgargs in "C2-table" "C2-table-large" "C2-table-stack" "C2-notebook" "C2 --num-points 1024 --n-gp 128"; do
  echo "$cfgargs"
  for rep in 1 2; do
    timeout 300 python bench.py --config $cfgargs --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
    SL_LIB_PATH=$P timeout 300 python bench.py --config $cfgargs --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
  done
done
} | tee gpurun_out/r04_gp_small_tickets.txt
