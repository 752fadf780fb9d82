#!/bin/bash
# Round 4, second GPU call: the suite on the device-resident level-set path + implicit V, then the
# HBM-bound lines and the headline quickly (three steps).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r04_pytest_gpu_2.log 2>&1
tail -15 gpurun_out/r04_pytest_gpu_2.log
for cfg in C4-lin C4-det C1 C2; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_2.jsonl
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_2.jsonl
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_2.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config']['name'], 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], 'finalize %s'%r.get('finalize_ms'), r['kernel'][:40], 'frac %.3f'%r['frac'], r.get('step_frac'), r.get('values_implicit'))
PY
