#!/bin/bash
# k_gp_small with 12 wavefronts per workgroup (three per SIMD) against 8: A/B in one call + the GP tests
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=gpurun_out/r04_gp_small_waves_ab.txt; : > $OUT
for rep in 1 2; do
  for w in 8 12; do
    for cfg in C2-table C2-table-large C2-table-stack; do
      SL_GP_SMALL_WAVES=$w timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | \
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('waves $w', '$cfg', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'], d['config'].get('safe_cells'))" >> $OUT
    done
  done
done
cat $OUT
timeout 900 python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_reference_gp.py tests/test_gpu_reference_safe_sets.py tests/test_gpu_configs.py tests/test_gpu_active_learning.py -q -x 2>&1 | tail -5
