#!/bin/bash
# Round 4, first GPU call: the full GPU suite on the shipped library (exclusion limits surveyed, not
# asserted), then the C5 development builds (tools/validate_dma.sh / tools/try_overlap.sh folded in).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export SL_EXCLUSION_SOFT=1
timeout 900 python -m pytest tests -m gpu -q --durations=20 > gpurun_out/r04_pytest_gpu_1.log 2>&1
tail -5 gpurun_out/r04_pytest_gpu_1.log
cp gpurun_out/parity_exclusions.json gpurun_out/r04_parity_exclusions_1.json 2>/dev/null
OUT=gpurun_out/r04_c5_variants.txt; : > $OUT
for lib in dma waves4 overlap overlap4; do
  [ -f safe_learning_amd/libslhip_$lib.so ] || continue
  echo "== $lib: tests" >> $OUT
  SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$lib.so timeout 600 python -m pytest tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py -x -q 2>&1 | tail -2 >> $OUT
done
line() {   # library, label, extra environment
  env SL_LIB_PATH=$PWD/safe_learning_amd/$1 $3 timeout 300 python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 14 2>/dev/null | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$2', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $OUT
}
for rep in 1 2; do
  line libslhip.so shipped ""
  line libslhip_dma.so "direct chunk copy" ""
  line libslhip_waves4.so "four wavefronts, one stream" ""
  line libslhip_overlap.so "four wavefronts + lookup beside it" ""
  line libslhip_overlap.so "overlap build, second stream off" "SL_BELLMAN4_OVERLAP=0"
  line libslhip_overlap4.so "overlap, lookup at 128 registers" ""
done
cat $OUT
