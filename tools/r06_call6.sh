#!/bin/bash
# Round 6, GPU call 6: whole suite after the miss-list / masked-q / flip-log changes, C5 lines, the
# driver's default bench command with the all-cores CPU baseline.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/call6_pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -6 $OUT/call6_pytest_all.log
timeout -k 5 600 python bench.py --config C5 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call6_c5.log 2>&1
echo "C5 rc=$?"; tail -1 $OUT/call6_c5.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['first_sweep_ms'], d['config']['time_to_convergence_s'])"
timeout -k 5 600 python bench.py --config C5-policy --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call6_c5p.log 2>&1
echo "C5-policy rc=$?"; tail -1 $OUT/call6_c5p.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])"
(time timeout -k 5 900 python bench.py) > $OUT/call6_bench_default.log 2>&1
echo "default bench rc=$?"; tail -5 $OUT/call6_bench_default.log | cut -c1-3000
