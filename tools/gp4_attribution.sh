#!/bin/bash
# Timing attribution of k_gp_sweep4 on the headline workload (results are garbage for skip != 0).
for skip in 0 1 2 3 4 7 8 15; do
  echo "SL_GP4_SKIP=$skip"
  SL_GP4_SKIP=$skip timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${1:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ms_per_step %.1f  kernel_ms %.1f  TF %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved']))
"
done
