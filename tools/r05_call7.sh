#!/bin/bash
# Round 5, GPU call 7 (the counter pass TCC_EA0_RDREQ_* of call 6 aborted inside rocprofv3 and then
# hung until the call's limit: every rocprofv3 below runs under `timeout`).  Vector-ALU counters of
# the VALU-bound kernels -> profiles/pmc_valu.json, matrix-pipe counters of k_gp_sweep4 at 48^4, one
# bench line per configuration, the driver's command, the smoke test, the GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r05_prof; O=gpurun_out/r05_final
mkdir -p $OUT $O
db() { find $1 -name "*_results.db" | head -1; }
for spec in "C4-lin C4-lin k_det_rows 1" "C4-det C4-det k_det_sweep 1" "C2-table-large C2-table-large k_gp_small 1" "C2-table-det C2-table-det k_det 0" \
            "C2-table C2-table k_gp_small 0" "C2-table-stack C2-table-stack k_gp_small 0" "C2-notebook C2-notebook k_gp_small 0" "C5-lookup C5 k_bellman_lookup 0"; do
  set -- $spec; key=$1; cfg=$2; sub=$3; second=$4
  extra=""; [ $cfg = C5 ] && extra="--max-sweeps 12"
  timeout -k 5 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES \
      -d $OUT/v_$key -o p -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $extra > $OUT/v_$key.log 2>&1
  q=""
  if [ $second = 1 ]; then
    timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
        -d $OUT/w_$key -o q -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $extra > $OUT/w_$key.log 2>&1
    q=$(db $OUT/w_$key)
  fi
  python tools/pmc_valu.py $key $sub $(db $OUT/v_$key) $q > $OUT/valu_$key.json 2>&1
  tail -c 300 $OUT/valu_$key.json | tr '\n' ' '; echo
  rm -rf $OUT/v_$key $OUT/w_$key
done
cp profiles/pmc_valu.json $O/pmc_valu.json
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    -d $OUT/pmc_a -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --num-points 48 > $OUT/pmc_a.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $OUT/pmc_a -name "*_results.db") > $OUT/r05_pmc_48.txt 2>&1
cat $OUT/r05_pmc_48.txt; rm -rf $OUT/pmc_a
bash tools/bench_configs.sh > $O/configs.txt 2>&1
cp gpurun_out/configs.jsonl $O/r05_configs.jsonl
cut -c1-210 $O/configs.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_command.log 2>$O/bench.err
cut -c1-1500 $O/r05_bench_driver_command.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke.log 2>&1
tail -2 $O/r05_smoke.log
timeout 400 python -m pytest tests -m gpu -q -x > $O/r05_pytest_gpu.log 2>&1
tail -4 $O/r05_pytest_gpu.log
