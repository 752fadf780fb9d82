#!/bin/bash
# Round 4, call 18: the whole GPU suite on the final k_gp_sweep4, its profiles (kernel trace,
# fabric traffic at 128^4, matrix-pipe counters at 48^4), the GP configurations, the driver's command.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_pytest_gpu_18.log 2>&1
tail -8 gpurun_out/r04_pytest_gpu_18.log
OUT=gpurun_out/r04_prof; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
python tools/kernel_stats.py $(find $OUT/trace -name "*_results.db" | head -1) > $OUT/r04_kernel_stats.md 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*_results.db" | head -1) $(find $OUT/pmc_write -name "*_results.db" | head -1) > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_fetch $OUT/pmc_write -name "*_results.db") > $OUT/r04_pmc_128.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    -d $OUT/pmc_a -o p -- $B --num-points 48 > $OUT/pmc_a.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_a -name "*_results.db") > $OUT/r04_pmc_48.txt 2>&1
grep -h '^{' $OUT/trace.log | cut -c1-600 > $OUT/r04_bench_lines.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_a
cat $OUT/r04_kernel_stats.md $OUT/r04_pmc_128.txt $OUT/r04_pmc_48.txt; cut -c1-300 $OUT/pmc_traffic.log
rm -f gpurun_out/r04_lines_18.jsonl
for cfg in C2 C3 C2-table-stack C2-notebook; do
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r04_lines_18.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_lines_18.jsonl'):
    d=json.loads(line); r=d['roofline']
    print(d['config'].get('name'), 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], r['kernel'][:60], 'frac %.3f'%r['frac'])
PY
timeout 900 python bench.py > gpurun_out/r04_bench_default_18.log 2>&1
tail -c 1500 gpurun_out/r04_bench_default_18.log
