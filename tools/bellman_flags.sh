#!/bin/bash
# k_bellman_mfma attribution (SL_BM_FLAGS, sl_bellman.hip): ms per C5 sweep under each setting,
# interleaved twice so that clock drift shows.  Output: gpurun_out/r02_bellman_flags.txt
mkdir -p gpurun_out
OUT=gpurun_out/r02_bellman_flags.txt
: > $OUT
run() {
  ms=$(SL_BM_FLAGS=$1 timeout 300 python bench.py --config C5 --steps 10 --warmup 2 --max-sweeps 12 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['roofline']['kernel_ms'])")
  echo "flags=$1 ($2): kernel_ms $ms" | tee -a $OUT
}
for rep in 1 2; do
  run 0 "wavefront-local ordering of the LDS stages (shipped)"
  run 4 "workgroup barriers"
  run 5 "no GEMM (epilogue only), barriers"
  run 6 "no epilogue (GEMM only), barriers"
done
