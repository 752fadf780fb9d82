#!/bin/bash
# Round 5, GPU call 4: the whole GPU suite, then C2 (static tile list, c_max on demand, no identity
# folds) against round 4's library and configuration lines for the 193..256-point route.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_call4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -8 $O/pytest.log
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.4f kernel_ms %.4f frac %.4f %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'][:50]))
if not ok: print('  failed')
"; }
{
for rep in 1 2; do
  echo "C2 tree"; timeout 200 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo "C2 tree, counter"; SL_GP4_TICKETS=1 timeout 200 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo "C2 prev kernels"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_prev.so timeout 200 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line
done
for n in 200 224 256; do
  echo "C2 1024^2, $n points: one panel of k_gp_sweep4"; timeout 300 python bench.py --config C2 --num-points 1024 --n-gp $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
  echo "C2 1024^2, $n points: k_gp_small"; SL_GP4_ONE_PANEL=0 timeout 300 python bench.py --config C2 --num-points 1024 --n-gp $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line
done
} | tee $O/c2.txt
