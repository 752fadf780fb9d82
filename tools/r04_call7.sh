#!/bin/bash
# Round 4, call 7: the extended two-rank test; k_gp_sweep4 diagnostics (phase attribution with
# SL_GP4_SKIP under the two-workgroup shape, instruction-cache / LDS / issue counters at 48^4) and
# the A/B of the DPP-rotated k_x fragments (libslhip_dpp.so, tools/build_variant.sh).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_call7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -x > $O/pytest_dist.log 2>&1
tail -5 $O/pytest_dist.log
if [ -f safe_learning_amd/libslhip_dpp.so ]; then
  SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_dpp.so timeout 900 python -m pytest tests/test_gpu_reference_gp.py tests/test_gpu_lyapunov.py -q -x > $O/pytest_dpp.log 2>&1
  tail -5 $O/pytest_dpp.log
fi
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  ms_per_step %.2f  kernel_ms %.2f  frac %.4f  %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'][:40]))
"; }
B="python bench.py --num-points 64 --steps 6 --warmup 2 --no-cpu-baseline"
for rep in 1 2 3; do
  echo "shipped"; timeout 200 $B 2>/dev/null | line
  if [ -f safe_learning_amd/libslhip_dpp.so ]; then
    echo "dpp"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_dpp.so timeout 200 $B 2>/dev/null | line
  fi
done | tee $O/ab_dpp.txt
for skip in 0 1 2 3 4 7 8 15; do
  echo "SL_GP4_SKIP=$skip"; SL_GP4_SKIP=$skip timeout 200 $B 2>/dev/null | line
done | tee $O/attribution.txt
P="python bench.py --num-points 48 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES \
    -d $O/pmc_i -o p -- $P > $O/pmc_i.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES \
    -d $O/pmc_l -o p -- $P > $O/pmc_l.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 \
    -d $O/pmc_v -o p -- $P > $O/pmc_v.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    -d $O/pmc_w -o p -- $P > $O/pmc_w.log 2>&1
python tools/pmc_dump.py k_gp_sweep $(find $O/pmc_i $O/pmc_l $O/pmc_v $O/pmc_w -name "*_results.db") > $O/pmc_diag_48.txt 2>&1
rm -rf $O/pmc_i $O/pmc_l $O/pmc_v $O/pmc_w
cat $O/pmc_diag_48.txt
