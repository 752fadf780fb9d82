#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r04_pytest_gpu_4.log 2>&1
tail -12 gpurun_out/r04_pytest_gpu_4.log
timeout 1500 bash tools/profile_r04.sh > gpurun_out/r04_profile.log 2>&1
tail -80 gpurun_out/r04_profile.log
