#!/bin/bash
# Round 4, call 26: seeds with non-temporal loads / stores (libslhip_nt.so): time at 64^4 and the
# fabric traffic of the headline launch, each against the library in the tree.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04_call26; rm -rf $O; mkdir -p $O
VL=$PWD/safe_learning_amd/libslhip_nt.so
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  kernel_ms %.2f frac %.4f' % (r['kernel_ms'], r['frac']))
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
{
for rep in 1 2 3; do
  echo "tree"; timeout 200 $B 2>/dev/null | line
  echo "nt"; SL_LIB_PATH=$VL timeout 200 $B 2>/dev/null | line
done
} | tee $O/ab.txt
for v in tree nt; do
  if [ $v = nt ]; then export SL_LIB_PATH=$VL; fi
  rocprofv3 --pmc FETCH_SIZE -d $O/f_$v -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/f_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/w_$v -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/w_$v.log 2>&1
  echo "$v" >> $O/traffic.txt
  python tools/pmc_dump.py k_gp_sweep $(find $O/f_$v $O/w_$v -name "*_results.db") >> $O/traffic.txt 2>&1
  rm -rf $O/f_$v $O/w_$v
done
cat $O/traffic.txt
