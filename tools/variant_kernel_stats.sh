#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a bench.py configuration under several
# development libraries (tools/build_variant.sh).   tools/variant_kernel_stats.sh "<bench flags>" <name> [<name> ...]
# -> gpurun_out/variant_kernel_stats.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
flags=$1; shift
out=gpurun_out/variant_kernel_stats.txt; : > $out
for v in "$@"; do
  d=/tmp/vks_$v; rm -rf $d
  SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $d -o t -- \
      python bench.py $flags --diagnostic --no-cpu-baseline > /tmp/vks_$v.log 2>&1
  echo "== $v: bench.py $flags" >> $out
  python tools/kernel_stats.py $(find $d -name "*_results.db" | head -1) 2>&1 | head -12 >> $out
done
cat $out
