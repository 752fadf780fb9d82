#!/bin/bash
# Round 6, GPU call 4: does the L2 miss rate of the factor's fragments cost k_gp_sweep4 time?  A
# development build whose row blocks all alias the first 1 MB of the factor (results meaningless)
# against the same build without the alias, 64^4; and the phases left out one by one (SL_GP4_SKIP).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
B="python bench.py --diagnostic --num-points 64 --steps 4 --warmup 1 --no-cpu-baseline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["roofline"]["kernel_ms"])'
for v in base alias; do
  for rep in 1 2; do
    echo -n "$v run $rep: "; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so timeout -k 5 200 $B 2>/dev/null | python -c "$pick"
  done
done
for skip in 1 2 8 9; do
  echo -n "base SL_GP4_SKIP=$skip: "; SL_GP4_SKIP=$skip SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_base.so timeout -k 5 200 $B 2>/dev/null | python -c "$pick"
done
echo -n "alias SL_GP4_SKIP=1: "; SL_GP4_SKIP=1 SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_alias.so timeout -k 5 200 $B 2>/dev/null | python -c "$pick"
