#!/bin/bash
# Round 5, GPU call 9: the GPU suite and the smoke test on the final tree.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/r05_pytest_gpu.log 2>&1
tail -6 $O/r05_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke.log 2>&1
tail -2 $O/r05_smoke.log
timeout 300 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-400
