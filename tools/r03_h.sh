#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_full_size.py tests/test_gpu_distributed.py -m gpu -x -q -k "row_kernel or deterministic or linear_dynamics or two_ranks or can_shrink or quirks or ties" > gpurun_out/r03_pytest_h.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r03_pytest_h.log
for e in "SL_DET_ROWS=1" "SL_DET_ROWS=0"; do
  echo "== $e"; env $e python bench.py --config C4-lin --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])"
done
echo "== C5 (margin 0)"; python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 13 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])"
