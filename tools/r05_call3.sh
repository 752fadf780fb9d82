#!/bin/bash
# Round 5, GPU call 3: the whole GPU suite on the tree, then round 4's kernel (prev), the diagonal
# split alone (diag) and the tree (diagonal split + training inputs / alpha' / seeds through buffer
# resources: no per-tile scratch stores) alternating at 64^4, and each once at 128^4.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_call3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f kernel_ms %.2f frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))
if not ok: print('  failed')
"; }
lib() { case $1 in tree) echo $PWD/safe_learning_amd/libslhip.so;; *) echo $PWD/safe_learning_amd/libslhip_$1.so;; esac; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
for rep in 1 2; do for v in prev diag tree; do
  echo "$v"; SL_LIB_PATH=$(lib $v) timeout 200 $B 2>/dev/null | line
done; done
for v in tree diag prev; do
  echo "$v, 128^4"; SL_LIB_PATH=$(lib $v) timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line
done
} | tee $O/ab.txt
