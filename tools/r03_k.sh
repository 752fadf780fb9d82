#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
B="python bench.py --num-points 64 --steps 3 --warmup 1 --no-cpu-baseline"
: > gpurun_out/r03_k64.txt
for rep in 1 2; do
  for v in p0 p1 p3; do
    echo "== $v rep $rep" >> gpurun_out/r03_k64.txt
    SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['safe_cells'], d['config']['negative_cells'], d['roofline']['kernel'])" >> gpurun_out/r03_k64.txt
  done
done
cat gpurun_out/r03_k64.txt
