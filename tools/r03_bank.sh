#!/bin/bash
# round 3: the logs that get committed under profiles/ (full GPU test suite, default bench line as the
# driver runs it, every configuration, rocprofv3 stats + PMC passes of the shipped kernels)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r03_pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.log 2>&1
tail -c 2500 gpurun_out/r03_bench_default.log
bash tools/bench_configs.sh 2>&1 | tail -12
bash tools/profile_r03.sh > gpurun_out/r03_prof.log 2>&1
tail -30 gpurun_out/r03_prof.log
bash tools/profile_r03_extra.sh > gpurun_out/r03_extra.log 2>&1
tail -30 gpurun_out/r03_extra.log
