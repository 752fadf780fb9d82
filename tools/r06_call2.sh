#!/bin/bash
# Round 6, GPU call 2: successor cache with the miss list - tests, C5-policy line, whole GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_successor_cache.py tests/test_gpu_rl.py tests/test_gpu_reference_policy_iteration.py -q -m gpu > $OUT/call2_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/call2_pytest.log
timeout -k 5 600 python bench.py --config C5-policy --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call2_c5p.log 2>&1
echo "C5-policy rc=$?"; tail -1 $OUT/call2_c5p.log
timeout -k 5 600 python bench.py --config C5-policy --steps 20 --warmup 2 --no-cpu-baseline --successor-cache off > $OUT/call2_c5p_off.log 2>&1
echo "C5-policy off rc=$?"; tail -1 $OUT/call2_c5p_off.log | cut -c1-400
timeout -k 5 1200 python -m pytest tests -q -m gpu -x > $OUT/call2_pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -8 $OUT/call2_pytest_all.log
