#!/bin/bash
# Round 4: the driver's command, every BASELINE configuration, and the CPU leg on SURVEY 8d's 2^24 cells.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.log 2>&1
tail -c 3000 gpurun_out/r04_bench_default.log
bash tools/bench_configs.sh > gpurun_out/r04_configs.log 2>&1
cp gpurun_out/configs.jsonl gpurun_out/r04_configs.jsonl 2>/dev/null
python - <<'PY'
import json
for line in open('gpurun_out/r04_configs.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config']['name'], 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], 'fin %s'%r.get('finalize_ms'), r['kernel'][:36], 'frac %.3f'%r['frac'], '%.3g %s'%(d['value'], d['unit']), d['config'].get('safe_cells'), d['config'].get('sweeps_to_convergence'))
PY
python bench.py --steps 1 --warmup 1 --cpu-cells 16777216 > gpurun_out/r04_cpu_baseline.log 2>&1
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/r04_cpu_baseline.log') if l.startswith('{')][-1]
print(json.dumps(d['cpu_baseline'], indent=1))
PY
