#!/bin/bash
# Round 4, final check of the tree: smoke(), the whole GPU suite, the driver's bench command.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r04_smoke_final.log 2>&1
tail -3 gpurun_out/r04_smoke_final.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_pytest_gpu_final.log 2>&1
tail -8 gpurun_out/r04_pytest_gpu_final.log
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r04_bench_final.log 2>&1
grep '^{' gpurun_out/r04_bench_final.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:d[k] for k in ('metric','value','unit','ms_per_step','steps','warmup','n_gpus','scaling','dtype','vs_baseline')})
print('roofline', r['frac'], r['achieved'], r['traffic'], r['kernel']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])"
