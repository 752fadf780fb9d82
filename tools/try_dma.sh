#!/bin/bash
# k_bellman4s with the chunk copy as direct global -> LDS loads (development build libslhip_dma.so):
# correctness of everything that runs the kernel, then the A/B on the C5 bench line
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/try_dma.txt
: > $OUT
DMA=$PWD/safe_learning_amd/libslhip_dma.so
SL_LIB_PATH=$DMA timeout 25 python -m pytest tests/test_gpu_rl.py -x -q -k "4x4x4 or ragged or sub_ranges" 2>&1 | tail -3 >> $OUT
for lib in $PWD/safe_learning_amd/libslhip.so $DMA; do
  SL_LIB_PATH=$lib timeout 15 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline --max-sweeps 6 2>/dev/null | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $OUT
done
cat $OUT
SL_LIB_PATH=$DMA timeout 60 python -m pytest tests/test_gpu_reference_policy_iteration.py tests/test_gpu_rl.py -x -q 2>&1 | tail -3 >> $OUT
tail -3 $OUT
