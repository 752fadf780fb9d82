#!/bin/bash
# Round 4, call 25: table flavours on large training sets (action table + k_gp_sweep4 + check):
# parity, then the notebook's grid with 512 training points on the old and the new route.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lyapunov.py -q -x -k "table" > gpurun_out/r04_pytest_25.log 2>&1
tail -12 gpurun_out/r04_pytest_25.log
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.3f kernel_ms %.3f  %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel'][:110]))
if not ok: print('  failed')
"; }
{
echo "C2-table-large, 512 training points (three passes)"
timeout 300 python bench.py --config C2-table-large --n-gp 512 --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/r04_err_25.txt | line
echo "the same on k_gp_sweep (SL_GP_CFG=3)"
SL_GP_CFG=3 timeout 300 python bench.py --config C2-table-large --n-gp 512 --steps 10 --warmup 2 --no-cpu-baseline 2>>gpurun_out/r04_err_25.txt | line
echo "C2-table-stack, 512 training points per head (three passes)"
timeout 300 python bench.py --config C2-table-stack --n-gp 512 --steps 10 --warmup 2 --no-cpu-baseline 2>>gpurun_out/r04_err_25.txt | line
echo "the same on k_gp_sweep (SL_GP_CFG=3)"
SL_GP_CFG=3 timeout 300 python bench.py --config C2-table-stack --n-gp 512 --steps 10 --warmup 2 --no-cpu-baseline 2>>gpurun_out/r04_err_25.txt | line
} | tee gpurun_out/r04_three_pass.txt
tail -3 gpurun_out/r04_err_25.txt
