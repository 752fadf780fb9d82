#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_full_size.py -m gpu -x -q -k "gp or GP or lengthscale" > gpurun_out/r03_pytest_e.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r03_pytest_e.log
B="python bench.py --num-points 64 --steps 3 --warmup 1 --no-cpu-baseline"
: > gpurun_out/r03_e64.txt
for rep in 1 2; do
  for v in new old; do
    lib=safe_learning_amd/libslhip_$v.so; [ $v = new ] && lib=safe_learning_amd/libslhip.so
    for skip in 0 16; do
      [ $v = old ] && [ $skip = 16 ] && continue
      echo "== $v skip $skip rep $rep" >> gpurun_out/r03_e64.txt
      SL_GP4_SKIP=$skip SL_LIB_PATH=$PWD/$lib $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])" >> gpurun_out/r03_e64.txt
    done
  done
done
cat gpurun_out/r03_e64.txt
