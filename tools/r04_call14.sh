#!/bin/bash
# Round 4, call 14: tiles with kinks - run constants once per tile, seeds per run.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call14; mkdir -p $O
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
for rep in 1 2; do
  echo "tree"; timeout 200 $B 2>/dev/null | line
  echo "no seeds"; SL_GP4_SEEDS=0 timeout 200 $B 2>/dev/null | line
done
} | tee $O/ab.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line | tee -a $O/ab.txt
P="python bench.py --num-points 48 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU \
    -d $O/pmc_v -o p -- $P > $O/pmc_v.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $O/pmc_v -name "*_results.db") > $O/pmc_valu_48.txt 2>&1
rm -rf $O/pmc_v
cat $O/pmc_valu_48.txt
