#!/bin/bash
# Round 4, third GPU call: new kernels first (adaptive primitives, sample glue, implicit V), then the
# whole suite, then the HBM-bound lines.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_active_learning.py tests/test_gpu_reference_safe_sets.py -q -x > gpurun_out/r04_pytest_gpu_3a.log 2>&1
tail -15 gpurun_out/r04_pytest_gpu_3a.log
timeout 1200 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r04_pytest_gpu_3.log 2>&1
tail -25 gpurun_out/r04_pytest_gpu_3.log
rm -f gpurun_out/r04_lines_3.jsonl
for cfg in C4-lin C4-det; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_3.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_3.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config']['name'], 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], 'finalize %s'%r.get('finalize_ms'), r['kernel'][:40], 'frac %.3f'%r['frac'], r.get('step_frac'), r.get('values_implicit'))
PY
