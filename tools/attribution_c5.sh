#!/bin/bash
# What each phase of k_bellman4s costs: the C5 bench line with development builds that leave phases
# out (tools/build_dev.py skipN -DSL_B4S_SKIP=N; results are wrong in those builds, times are not)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/attribution_c5.txt
: > $OUT
for lib in libslhip.so $(cd safe_learning_amd; ls libslhip_skip*.so | sort -t p -k3 -n) libslhip.so; do
  SL_LIB_PATH=$PWD/safe_learning_amd/$lib python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline --max-sweeps 6 2>/dev/null | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $OUT
done
cat $OUT
