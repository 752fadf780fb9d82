#!/bin/bash
# Round 6: records on the sources with the span-bounded streaming pass and the library's own kernel
# timing: counter passes of the VALU-bound kernels (profiles/pmc_valu.json is tied to the sources by
# sha256), the GPU suite's log, one line per configuration, the default bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_final5; mkdir -p $OUT
timeout -k 5 300 python tools/step_probe.py C1 3000 2>&1 | grep "us/step" > $OUT/step_probe.txt; cat $OUT/step_probe.txt
bash tools/profile_r06.sh valu > $OUT/profile.log 2>&1
echo "valu rc=$?"
cp profiles/pmc_valu.json $OUT/pmc_valu.json
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r06_pytest_gpu.log
cp gpurun_out/parity_exclusions.json $OUT/r06_parity_exclusions.json 2>/dev/null
bash tools/bench_configs.sh > $OUT/r06_configs_table.txt 2>&1
cp gpurun_out/configs.jsonl $OUT/r06_configs.jsonl
cat $OUT/r06_configs_table.txt
