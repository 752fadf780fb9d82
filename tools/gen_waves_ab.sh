#!/bin/bash
# A/B of the occupancy bound of the GENERAL (table V / interpolated policy) per-cell kernels of
# sl_kernels.hip: __launch_bounds__(256, SL_GEN_WAVES) with 1 (256 + 128..240 registers, one wavefront per
# SIMD) against 2 (256 registers + scratch, two per SIMD).  Libraries: tools/build_variant.sh gen<k> sl_kernels
# sl_kernels.hip -DSL_GEN_WAVES=<k>.  Writes gpurun_out/gen_waves_ab.txt.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/gen_waves_ab.txt; : > $out
for v in gen1 gen2 gen1 gen2; do
  for cfg in "C2-table-det" "C2-table-large --n-gp 512"; do
    d=/tmp/prof_${v}_$RANDOM
    SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so rocprofv3 --kernel-trace --stats -d $d -o p -- \
      python bench.py --config $cfg --diagnostic --no-cpu-baseline --steps 20 --warmup 3 > /tmp/line.txt 2>/tmp/err.txt
    python - "$v" "$cfg" $d >> $out <<'PY'
import sys, json, glob, csv
v, cfg, d = sys.argv[1:4]
line = [l for l in open('/tmp/line.txt') if l.startswith('{')]
ms = json.loads(line[-1])['ms_per_step'] if line else None
print("[%s] %s ms_per_step %s" % (v, cfg, ms))
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.5:
            print("    %-70s calls %s avg %.4f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e6))
PY
  done
done
cat $out
