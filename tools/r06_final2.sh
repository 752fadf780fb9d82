#!/bin/bash
# Round 6, last GPU call: test suite, smoke, the driver's bench command and one line per configuration
# on the tree as committed (the counter passes of tools/r06_final.sh stay valid: no kernel source changed).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_final2; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r06_pytest_gpu.log
cp gpurun_out/parity_exclusions.json $OUT/r06_parity_exclusions.json 2>/dev/null
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 $OUT/r06_smoke.log
(time timeout -k 5 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/r06_bench_driver_command.log 2>&1
echo "driver bench rc=$?"; grep real $OUT/r06_bench_driver_command.log
timeout -k 5 900 python bench.py > $OUT/r06_bench_default.log 2>&1
echo "default bench rc=$?"
bash tools/bench_configs.sh > $OUT/r06_configs_table.txt 2>&1
cp gpurun_out/configs.jsonl $OUT/r06_configs.jsonl
cat $OUT/r06_configs_table.txt
