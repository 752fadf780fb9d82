#!/bin/bash
# Round 6, GPU call 7: cached sweeps with several entries in flight, no per-sweep syncs; RL tests.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_successor_cache.py tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py tests/test_gpu_distributed.py tests/test_gpu_notebook_loop.py -q -m gpu > $OUT/call7_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/call7_pytest.log
for cfg in C5 C5-policy; do
timeout -k 5 600 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call7_$cfg.log 2>&1
echo "$cfg rc=$?"; tail -1 $OUT/call7_$cfg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('first_sweep_ms'), d['config'].get('time_to_convergence_s'))"
done
