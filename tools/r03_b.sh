#!/bin/bash
# round 3, GPU call B: GPU tests, k_gp_sweep4 with fillers vs the round-2 kernel at 64^4, default bench
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_b.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r03_pytest_b.log
B="python bench.py --num-points 64 --steps 3 --warmup 1 --no-cpu-baseline"
: > gpurun_out/r03_b64.txt
for rep in 1 2; do
  for v in old new; do
    lib=safe_learning_amd/libslhip_$v.so; [ $v = new ] && lib=safe_learning_amd/libslhip.so
    echo "== $v rep $rep" >> gpurun_out/r03_b64.txt
    SL_LIB_PATH=$PWD/$lib $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['config']['c_max'])" >> gpurun_out/r03_b64.txt
  done
done
for skip in 1 2 8 15; do
  echo "== new skip $skip (slow path)" >> gpurun_out/r03_b64.txt
  SL_GP4_SKIP=$skip $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r03_b64.txt
done
cat gpurun_out/r03_b64.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_default_b.log 2>&1
tail -c 1500 gpurun_out/r03_bench_default_b.log
