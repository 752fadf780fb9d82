#!/bin/bash
# Round 4, call 32: where k_gp_sweep4 (256-row panels: training sets padded to 256) overtakes
# k_gp_small (a wavefront per tile, 16x16x4 MFMAs, one exponential per point and cell).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('    kernel_ms %.3f  %s' % (r['kernel_ms'], r['kernel'][:60]))
if not ok: print('    failed')
"; }
{
for n in 96 128 160 192 224 256; do
  echo "C2-table-large (2001 x 1501, table V + table policy), $n training points"
  timeout 300 python bench.py --config C2-table-large --n-gp $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line
  SL_GP_CFG=2 timeout 300 python bench.py --config C2-table-large --n-gp $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line
  echo "C2 (1024^2, quadratic V, closed-form policy), $n training points"
  timeout 300 python bench.py --config C2 --num-points 1024 --n-gp $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line
  SL_GP_CFG=2 timeout 300 python bench.py --config C2 --num-points 1024 --n-gp $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line
done
} | tee gpurun_out/r04_small_vs_gp4.txt
