#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_active_learning.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r03_pytest_f.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r03_pytest_f.log
: > gpurun_out/r03_f.txt
for cfg in C2-table-large C2-table; do
  for small in 1 0; do
    echo "== $cfg SL_GP_SMALL=$small" >> gpurun_out/r03_f.txt
    SL_GP_SMALL=$small python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])" >> gpurun_out/r03_f.txt
  done
done
cat gpurun_out/r03_f.txt
