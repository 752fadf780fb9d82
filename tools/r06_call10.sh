#!/bin/bash
# Round 6, GPU call 10: is C3 (k_gp_sweep4<d=2> with records + k_nn_check_mfma) slower than in round 5
# because of this round's clean-up of sl_gp4.hip, or because of the box?  The round-5 library
# (libslhip_r05.so, built from commit 04ebbe6) against the tree's on ONE box, alternating.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"])'
for rep in 1 2; do
  for cfg in C3 C4-lin C4-det; do
    echo -n "tree $cfg: "; timeout -k 5 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$pick"
    echo -n "r05  $cfg: "; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_r05.so timeout -k 5 300 python bench.py --diagnostic --config $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$pick"
  done
done
timeout -k 5 600 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -3
