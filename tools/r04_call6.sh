#!/bin/bash
# Round 4: the region kernels, then the whole suite on the final tree, the two table lines with the
# final k_gp_small, and the traffic counters on the (instruction-identical) final sl_gp4.hip.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_regions.py -q -x > gpurun_out/r04_pytest_gpu_6a.log 2>&1
tail -8 gpurun_out/r04_pytest_gpu_6a.log
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_pytest_gpu_6.log 2>&1
tail -8 gpurun_out/r04_pytest_gpu_6.log
rm -f gpurun_out/r04_lines_6.jsonl
for cfg in C2-table C2-table-large C2-table-stack; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_6.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_6.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config']['name'], 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], r['kernel'], 'frac %.3f'%r['frac'])
PY
OUT=gpurun_out/r04_prof; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*_results.db" | head -1) $(find $OUT/pmc_write -name "*_results.db" | head -1) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/r04_pmc_128.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/r04_pmc_128.txt; cut -c1-400 $OUT/pmc_traffic.log
