#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for v in b4eps0 b4vis main; do
  lib=safe_learning_amd/libslhip_$v.so; [ $v = main ] && lib=safe_learning_amd/libslhip.so
  echo "== $v"; SL_LIB_PATH=$PWD/$lib python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --max-sweeps 13 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])"
done
