#!/bin/bash
# Round 6, GPU call 5: (a) seeds of k_gp_sweep4 with the non-temporal cache policy - time, L2 hit rate,
# fabric bytes against the same development build without it (64^4); (b) kernel traces of C5 / C5-policy.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
B="python bench.py --diagnostic --num-points 64 --steps 3 --warmup 1 --no-cpu-baseline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["roofline"]["kernel_ms"])'
for v in base seednt; do
  export SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_$v.so
  for rep in 1 2; do echo -n "$v run $rep: "; timeout -k 5 200 $B 2>/dev/null | python -c "$pick"; done
  k=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    k=$((k+1))
    timeout -k 5 150 rocprofv3 --pmc $set -d $OUT/nt_${v}_$k -o p -- $B > $OUT/nt_${v}_$k.log 2>&1
    echo "$v pass $k ($set): rc=$?"
  done
  python tools/pmc_dump.py k_gp_sweep $(find $OUT/nt_${v}_* -name "*_results.db") > $OUT/r06_gp4_seednt_$v.txt 2>&1
  cat $OUT/r06_gp4_seednt_$v.txt
  rm -rf $OUT/nt_${v}_?
done
unset SL_LIB_PATH
bash tools/profile_r06.sh c5
cat gpurun_out/r06_prof/r06_C5_kernel_stats.md | head -30
cat gpurun_out/r06_prof/r06_C5-policy_kernel_stats.md | head -20
tail -1 gpurun_out/r06_prof/trace_c5.log | cut -c1-200
