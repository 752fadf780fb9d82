#!/bin/bash
# Round 6, GPU call 1: the successor cache - its tests, the RL tests around it, C5 / C5-policy lines.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_successor_cache.py tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py -x -q -m gpu > $OUT/call1_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/call1_pytest.log
timeout -k 5 600 python bench.py --config C5 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call1_c5.log 2>&1
echo "C5 rc=$?"; tail -3 $OUT/call1_c5.log
timeout -k 5 600 python bench.py --config C5-policy --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call1_c5p.log 2>&1
echo "C5-policy rc=$?"; tail -3 $OUT/call1_c5p.log
