#!/bin/bash
# A/B of two builds of sl_gp_small.hip (tools/build_variant.sh <name> sl_gp_small sl_gp_small.hip) on the
# configurations k_gp_small serves (CFGS='cfg a;cfg b': another list).   tools/gp_small_ab.sh <name A> <name B>   -> gpurun_out/gp_small_ab.txt
cd "$(dirname "$0")/.."
out=gpurun_out/gp_small_ab.txt; : > $out
for rep in 1 2; do
for v in "$@"; do
  IFS=';' read -ra CFG_LIST <<< "${CFGS:-C2-table;C2-table-large;C2-table-stack;C2-notebook;C2 --n-gp 128 --num-points 2048;C4 --num-points 48 --n-gp 192}"
  for cfg in "${CFG_LIST[@]}"; do
    SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so python bench.py --config $cfg --diagnostic --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('[$v] %-36s ms_per_step %8.4f kernel_ms %8.4f  %s' % ('$cfg', d['ms_per_step'], r['kernel_ms'], r['kernel'][:90]))" >> $out
  done
done
done
cat $out
