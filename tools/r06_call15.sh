#!/bin/bash
# Round 6, GPU call 15: the table flavours on small training sets as three passes (action table,
# lean fast-path k_gp_small writing posterior records, k_check_records) against the fused kernel.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
SL_GP_SMALL_SPLIT=1 timeout -k 5 1200 python -m pytest tests -q -m gpu -x -k "small or notebook or table or kernel or stack or c2" > $OUT/call15_tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/call15_tests.log
one() { timeout 300 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --diagnostic 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['config']['name'], round(d['ms_per_step'],3), round(r['kernel_ms'],3), r.get('kernel'), d['config'].get('safe_cells'))"; }
for c in C2-table C2-table-large C2-table-stack C2-notebook; do
  one $c fused
  SL_GP_SMALL_SPLIT=1 one $c split
  SL_GP_SMALL_SPLIT=1 SL_GP_SMALL_WAVES=8 one $c split-waves8
done
SL_GP_SMALL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/split_t -o t -- python bench.py --config C2-notebook --steps 6 --warmup 2 --no-cpu-baseline --diagnostic > /dev/null 2>&1
python tools/kernel_stats.py $(find $OUT/split_t -name "*_results.db" | head -1) | head -8
SL_GP_SMALL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/split_u -o t -- python bench.py --config C2-table-large --steps 6 --warmup 2 --no-cpu-baseline --diagnostic > /dev/null 2>&1
python tools/kernel_stats.py $(find $OUT/split_u -name "*_results.db" | head -1) | head -8
rm -rf $OUT/split_t $OUT/split_u
