#!/bin/bash
# Round 4, call 24: whole GPU suite after the kernel family; the notebook's model with its own kernels
# (C2-notebook) beside the RBF stack (C2-table-stack).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_pytest_gpu_24.log 2>&1
tail -8 gpurun_out/r04_pytest_gpu_24.log
rm -f gpurun_out/r04_lines_24.jsonl
for cfg in C2-table-stack C2-notebook; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r04_err_24.txt | grep '^{' >> gpurun_out/r04_lines_24.jsonl
  tail -2 gpurun_out/r04_err_24.txt
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_24.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config'].get('name'), 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], r['kernel'][:70], 'frac %.3f'%r['frac'], d['config'].get('safe_cells'), d['config'].get('negative_cells'))
PY
