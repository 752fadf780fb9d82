cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_extra
for c in C2-table-large C2-notebook C2-table-det C2 C3; do
  steps=20; [ $c = C3 ] && steps=3
  d=/tmp/tr_$c; rm -rf $d
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $d -o t -- python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline > /tmp/tr_$c.log 2>&1
  python tools/kernel_stats.py $(find $d -name "*_results.db" | head -1) > gpurun_out/r06_extra/r06_${c}_kernel_stats.md 2>&1
  head -6 gpurun_out/r06_extra/r06_${c}_kernel_stats.md
done
