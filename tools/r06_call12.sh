#!/bin/bash
# Round 6, GPU call 12: what does the LATENCY of the seed loads (and the LDS stores) of a regenerated
# chunk cost k_gp_sweep4?  Development builds (results meaningless), 64^4.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
B="python bench.py --diagnostic --num-points 64 --steps 4 --warmup 1 --no-cpu-baseline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["roofline"]["kernel_ms"])'
for rep in 1 2; do
for v in base noseedload noseedload_nostore; do
  echo -n "$v: "; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so timeout -k 5 200 $B 2>/dev/null | python -c "$pick"
done
done
