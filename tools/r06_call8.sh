#!/bin/bash
# Round 6, GPU call 8: two cells per thread in the cached max sweep.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_successor_cache.py tests/test_gpu_rl.py -q -m gpu -x > $OUT/call8_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/call8_pytest.log
timeout -k 5 600 python bench.py --config C5 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call8_C5.log 2>&1
echo "C5 rc=$?"; tail -1 $OUT/call8_C5.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('first_sweep_ms'), d['config'].get('time_to_convergence_s'))"
