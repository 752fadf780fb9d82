#!/bin/bash
# Round 5, GPU call 11: get_safe_sample over an action grid (device pairs), and the sampling tests.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_active_learning.py tests/test_gpu_notebook_loop.py tests/test_gpu_reference_safe_sets.py -q 2>&1 | tail -6
