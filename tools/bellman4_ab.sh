#!/bin/bash
# C5 Bellman max sweep: k_bellman4 (4x4x4) against k_bellman_mfma (16x16x4), interleaved, plus
# the attribution flags of k_bellman4 (SL_BM_FLAGS: 1 no GEMM, 2 no epilogue, 16 no generation,
# 64 one working wavefront per SIMD, 128 no value-table lookup).
mkdir -p gpurun_out
OUT=gpurun_out/r02_bellman4_ab.txt
: > $OUT
run() {
  ms=$(SL_BELLMAN4=$1 SL_BM_FLAGS=$2 SL_BELLMAN4_QUARTER=${4:-1} SL_BELLMAN4_SPLIT=${5:-1} timeout 300 python bench.py --config C5 --steps 10 --warmup 2 --max-sweeps 12 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))")
  echo "SL_BELLMAN4=$1 flags=$2 ($3): kernel_ms $ms" | tee -a $OUT
}
for rep in 1 2; do
  run 1 0 "k_bellman4 + k_bellman_lookup (split sweep)"
  run 1 0 "k_bellman4, fused epilogue" 1 0
  run 0 0 "k_bellman_mfma"
  run 1 2 "k_bellman4, no epilogue" 1 0
  run 1 18 "k_bellman4, GEMM only (no generation, no epilogue)" 1 0
done
