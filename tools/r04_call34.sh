#!/bin/bash
# Round 4, call 34: k_gp_small with the kernel family as a template flag (the RBF instantiation has
# no branch in its generation loop): parity of the GP sweeps, the table configurations.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_gp_kernels.py tests/test_gpu_reference_gp.py tests/test_gpu_notebook_loop.py tests/test_gpu_lyapunov.py -q -x 2>&1 | tail -2
rm -f gpurun_out/r04_lines_34.jsonl
for cfg in C2-table C2-table-large C2-table-stack C2-notebook; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_34.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_34.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config'].get('name'), 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], r['kernel'][:70], 'frac %.3f'%r['frac'])
PY
