#!/bin/bash
# Round 6: the records kept under profiles/ - GPU test suite, smoke, the driver's bench command, every
# BASELINE configuration, SURVEY 8d's literal GP variant, the host time of a small step, profiles of the
# shipped kernels.   gpurun --timeout 3000 -- 'bash tools/r06_final.sh'   then copy gpurun_out/r06_final/*
# into profiles/ and run tools/profiles_index.py.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r06_pytest_gpu.log
cp gpurun_out/parity_exclusions.json $OUT/r06_parity_exclusions.json 2>/dev/null
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 $OUT/r06_smoke.log
(time timeout -k 5 1200 python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/r06_bench_driver_command.log 2>&1
echo "driver bench rc=$?"; grep real $OUT/r06_bench_driver_command.log
timeout -k 5 900 python bench.py > $OUT/r06_bench_default.log 2>&1
echo "default bench rc=$?"
timeout -k 5 600 python bench.py --gp-variant survey --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r06_survey_variant.jsonl 2>/dev/null
echo "survey rc=$?"
timeout -k 5 300 python tools/step_probe.py C1 3000 2>&1 | grep "us/step" > $OUT/r06_step_host_time.txt; cat $OUT/r06_step_host_time.txt
bash tools/profile_r06.sh headline valu c5 > $OUT/profile.log 2>&1
echo "profiles rc=$?"
cp gpurun_out/r06_prof/*.md gpurun_out/r06_prof/*.txt gpurun_out/r06_prof/*.json $OUT/ 2>/dev/null
# the per-configuration lines AFTER the counter passes: bench.py reports `bound: "valu"` from profiles/pmc_valu.json,
# which the valu part above has just regenerated for the sources of this tree (a stale entry falls back to the
# matrix-pipe figure)
bash tools/bench_configs.sh > $OUT/r06_configs_table.txt 2>&1
cp gpurun_out/configs.jsonl $OUT/r06_configs.jsonl
cat $OUT/r06_configs_table.txt
ls $OUT
