#!/bin/bash
# Round 4, call 8: k_gp_sweep4 with the sequence seeds kept between the panels of a tile
# (SL_GP4_SEEDS=0 switches the reuse off: the round-3 behaviour) - parity, A/B at 64^4, headline.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py tests/test_gpu_configs.py -q -x > $O/pytest_gp4.log 2>&1
tail -5 $O/pytest_gp4.log
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f  kernel_ms %.2f  frac %.4f  %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'][:40]))
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
for rep in 1 2 3; do
  echo "seeds off"; SL_GP4_SEEDS=0 timeout 200 $B 2>/dev/null | line
  echo "seeds on"; timeout 200 $B 2>/dev/null | line
done | tee $O/ab_seeds.txt
for skip in 1 2 3; do
  echo "SL_GP4_SKIP=$skip"; SL_GP4_SKIP=$skip timeout 200 $B 2>/dev/null | line
done | tee $O/attribution.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_128.log 2>&1
grep '^{' $O/bench_128.log | line
timeout 300 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | line
