#!/bin/bash
# First GPU call of the next round: is the direct global -> LDS chunk copy of k_bellman4s
# (-DSL_B4S_DMA) ready to become the default?  Build the development library first (CPU):
#     python tools/build_dev.py dma -DSL_B4S_DMA
# then, on the GPU box:  bash tools/validate_dma.sh      ->  gpurun_out/validate_dma.txt
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/validate_dma.txt
mkdir -p gpurun_out; : > $OUT
DMA=$PWD/safe_learning_amd/libslhip_dma.so
[ -f $DMA ] || { echo "build libslhip_dma.so first" | tee -a $OUT; exit 1; }
# everything that launches the Bellman kernels, on the development build
SL_LIB_PATH=$DMA python -m pytest tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py \
    tests/test_gpu_distributed.py tests/test_gpu_bench.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3 >> $OUT
# A/B on the C5 line, alternating in one call
for rep in 1 2; do
  for lib in $PWD/safe_learning_amd/libslhip.so $DMA; do
    SL_LIB_PATH=$lib python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 14 2>/dev/null | grep '^{' | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $OUT
  done
done
cat $OUT
