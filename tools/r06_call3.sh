#!/bin/bash
# Round 6, GPU call 3: whole GPU suite on the cleaned-up library, C5 lines, shard balance, gp4 at 64^4.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/call3_pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -12 $OUT/call3_pytest_all.log
timeout -k 5 300 python bench.py --num-points 64 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/call3_gp4_64.log 2>&1
echo "gp4 64^4 rc=$?"; tail -1 $OUT/call3_gp4_64.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'])"
timeout -k 5 600 python bench.py --config C5 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/call3_c5.log 2>&1
echo "C5 rc=$?"; tail -1 $OUT/call3_c5.log | cut -c1-300
timeout -k 5 600 python tools/shard_balance.py --config C4 > $OUT/r06_shard_balance_C4.md 2>&1
echo "shard C4 rc=$?"; grep -E "balance|Whole" $OUT/r06_shard_balance_C4.md
timeout -k 5 600 python tools/shard_balance.py --config C5 --repeat 3 > $OUT/r06_shard_balance_C5.md 2>&1
echo "shard C5 rc=$?"; grep -E "balance|Whole" $OUT/r06_shard_balance_C5.md
