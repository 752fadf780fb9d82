#!/bin/bash
# Round 5, GPU call 2: which part of the diagonal-block split of k_gp_sweep4 is wrong / slow?
# variants (tools/build_variant.sh, -DSL_GP4_DIAG_*): NZ8 = never leaves the diagonal stream early,
# RELOAD = the blocks below re-read their first k_x fragments, LATE_A = their first A fragments are
# requested behind the diagonal stream.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_diag_bisect; mkdir -p $O
T="tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py"
for v in prev tree diag_NZ8 diag_RELOAD diag_LATE_A; do
  lib=$PWD/safe_learning_amd/libslhip_$v.so; [ $v = tree ] && lib=$PWD/safe_learning_amd/libslhip.so
  echo "== $v"
  SL_LIB_PATH=$lib timeout 300 python -m pytest $T -q -k "n512 or n1024 or gp4 or headline or seeds" 2>&1 | tail -4 | cut -c1-300
done 2>&1 | tee $O/pytest.txt
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f kernel_ms %.2f frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))
if not ok: print('  failed')
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
for v in prev diag_NZ8 diag_RELOAD diag_LATE_A tree prev; do
  lib=$PWD/safe_learning_amd/libslhip_$v.so; [ $v = tree ] && lib=$PWD/safe_learning_amd/libslhip.so
  echo "$v"; SL_LIB_PATH=$lib timeout 200 $B 2>/dev/null | line
done 2>&1 | tee $O/ab.txt
