#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_r4.so python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_full_size.py -m gpu -x -q -k "cartpole and (gp or GP or lengthscale)" > gpurun_out/r03_pytest_i.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r03_pytest_i.log
B="python bench.py --num-points 64 --steps 3 --warmup 1 --no-cpu-baseline"
: > gpurun_out/r03_i64.txt
for rep in 1 2; do
  for v in r4 new; do
    lib=safe_learning_amd/libslhip_$v.so; [ $v = new ] && lib=safe_learning_amd/libslhip.so
    echo "== $v rep $rep" >> gpurun_out/r03_i64.txt
    SL_LIB_PATH=$PWD/$lib $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])" >> gpurun_out/r03_i64.txt
  done
done
cat gpurun_out/r03_i64.txt
SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_r4.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128^4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'])"
