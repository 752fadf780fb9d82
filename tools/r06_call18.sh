#!/bin/bash
# Round 6, GPU call 18: k_gp_small reads the kernel description from LDS and evaluates a lane's two
# training points in one walk over the factors: parity tests of the kernel families, then timings.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 1200 python -m pytest tests -q -m gpu -x -k "small or notebook or table or kernel or stack or c2" > $OUT/call18_tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/call18_tests.log
one() { timeout 300 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['config']['name'], round(d['ms_per_step'],3), round(r['kernel_ms'],3), r.get('kernel'), d['config'].get('safe_cells'))"; }
for c in C2-table C2-table-large C2-table-stack C2-notebook; do one $c shipped; done
SL_GP_SMALL_WAVES=8 one C2-notebook waves8
