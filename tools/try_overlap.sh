#!/bin/bash
# Next round, prepared blind (no GPU time was left to run it): the GEMM half of the split Bellman
# sweep as ONE workgroup of four wavefronts per CU (256 registers each, chunk copy by direct
# global -> LDS loads) with the lookup of the previous round on a second stream beside it.
# Build on the CPU first:
#     python tools/build_dev.py waves4   -DSL_B4S_WAVES=4 -DSL_B4S_DMA
#     python tools/build_dev.py overlap  -DSL_B4S_WAVES=4 -DSL_B4S_DMA -DSL_B4S_OVERLAP
#     python tools/build_dev.py overlap4 -DSL_B4S_WAVES=4 -DSL_B4S_DMA -DSL_B4S_OVERLAP -DSL_B4_LOOKUP_BLOCKS=4
# then on the GPU box:  bash tools/try_overlap.sh   ->  gpurun_out/try_overlap.txt
# (SL_BELLMAN4_OVERLAP=0 switches the second stream off inside an overlap build.)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/try_overlap.txt
mkdir -p gpurun_out; : > $OUT
line() {   # library, label, extra environment
  env SL_LIB_PATH=$PWD/safe_learning_amd/$1 $3 python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 14 2>/dev/null | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$2', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'sweeps', d['config'].get('sweeps_to_convergence'))" >> $OUT
}
for lib in waves4 overlap overlap4; do
  [ -f safe_learning_amd/libslhip_$lib.so ] || continue
  echo "== $lib: tests" >> $OUT
  SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$lib.so python -m pytest tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py -x -q 2>&1 | tail -2 >> $OUT
done
for rep in 1 2; do
  line libslhip.so shipped ""
  [ -f safe_learning_amd/libslhip_waves4.so ] && line libslhip_waves4.so "four wavefronts, one stream" ""
  [ -f safe_learning_amd/libslhip_overlap.so ] && line libslhip_overlap.so "four wavefronts + lookup beside it" ""
  [ -f safe_learning_amd/libslhip_overlap.so ] && line libslhip_overlap.so "overlap build, second stream off" "SL_BELLMAN4_OVERLAP=0"
  [ -f safe_learning_amd/libslhip_overlap4.so ] && line libslhip_overlap4.so "overlap, lookup at 128 registers" ""
done
cat $OUT
