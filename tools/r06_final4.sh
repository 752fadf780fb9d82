#!/bin/bash
# Round 6: counter passes of the VALU-bound kernels on the final sources (profiles/pmc_valu.json is tied
# to them by sha256), the GPU suite's log and one line per configuration with those entries in place.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_final4; mkdir -p $OUT
bash tools/profile_r06.sh valu > $OUT/profile.log 2>&1
echo "valu rc=$?"
cp profiles/pmc_valu.json $OUT/pmc_valu.json
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r06_pytest_gpu.log
cp gpurun_out/parity_exclusions.json $OUT/r06_parity_exclusions.json 2>/dev/null
bash tools/bench_configs.sh > $OUT/r06_configs_table.txt 2>&1
cp gpurun_out/configs.jsonl $OUT/r06_configs.jsonl
cat $OUT/r06_configs_table.txt
