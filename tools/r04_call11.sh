#!/bin/bash
# Round 4, call 11: the non-MFMA phases of k_gp_sweep4 alone (SL_GP4_SKIP=8, one workgroup per CU)
# and inside the full kernel, for the diagnostic builds of tools/gp4_diag_variants.py.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call11; mkdir -p $O
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  kernel_ms %.2f' % r['kernel_ms'])
if not ok: print('  failed')
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
for v in base g_noload g_nostore g_noexp g_norec m_noload m_nonop nobarrier; do
  L=$PWD/safe_learning_amd/libslhip_$v.so
  [ -f $L ] || continue
  echo "$v: phases alone, one workgroup per CU"; SL_LIB_PATH=$L SL_GP4_SKIP=8 SL_GP4_WGS=1 timeout 200 $B 2>$O/err_$v.txt | line
  echo "$v: full kernel"; SL_LIB_PATH=$L timeout 200 $B 2>>$O/err_$v.txt | line
done
echo "tree: phases alone, skip 9 / 10 / 11 / 15, one workgroup per CU"
for skip in 9 10 11 15; do SL_GP4_SKIP=$skip SL_GP4_WGS=1 timeout 200 $B 2>/dev/null | line; done
echo "tree: full kernel"; timeout 200 $B 2>/dev/null | line
} | tee $O/phases.txt
tail -3 $O/err_g_noload.txt
