#!/bin/bash
# Round 4, call 40: the GP sweeps on the panels that hold training points; whole GPU suite; traffic
# of the headline launch re-measured for the final sl_gp4.hip; the driver's bench command.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu_final.log 2>&1
tail -3 gpurun_out/r04_pytest_gpu_final.log
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('    kernel_ms %.3f frac %.4f  %s' % (r['kernel_ms'], r['frac'], r['kernel'][:50]))
if not ok: print('    failed')
"; }
{
for n in 520 600 800 1024; do
  echo "cart-pole 64^4, $n training points"
  timeout 300 python bench.py --num-points 64 --n-gp $n --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | line
done
} | tee gpurun_out/r04_gp4_panels.txt
OUT=gpurun_out/r04_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*_results.db" | head -1) $(find $OUT/pmc_write -name "*_results.db" | head -1) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/r04_pmc_128.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/r04_pmc_128.txt
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r04_bench_final.log 2>&1
grep '^{' gpurun_out/r04_bench_final.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'])"
