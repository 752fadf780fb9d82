#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_r4.so python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_j.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r03_pytest_j.log
: > gpurun_out/r03_j.txt
for cfg in C3 C2; do
  for v in r4 new; do
    lib=safe_learning_amd/libslhip_$v.so; [ $v = new ] && lib=safe_learning_amd/libslhip.so
    steps=10; [ $cfg = C3 ] && steps=3
    echo "== $cfg $v" >> gpurun_out/r03_j.txt
    SL_LIB_PATH=$PWD/$lib python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])" >> gpurun_out/r03_j.txt
  done
done
cat gpurun_out/r03_j.txt
