#!/bin/bash
# Round 5, GPU call 5a: the refinement-carry tests, and k_bellman_lookup at 2 / 3 (shipped) / 4
# wavefronts per SIMD on the C5 line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_call5a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lyapunov.py tests/test_gpu_adaptive.py tests/test_gpu_reference_safe_sets.py tests/test_gpu_mask_state.py -q -k "refinement or adaptive or mask_state or closed_form" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
line() { python -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        ok=True; d=json.loads(l); r=d['roofline']; print('  ms_per_step %.3f kernel_ms %.3f  %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel'][:70]))
if not ok: print('  failed')
"; }
B="python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --max-sweeps 15"
{
for rep in 1 2; do
  echo "tree (3 wavefronts per SIMD)"; timeout 300 $B 2>/dev/null | line
  echo "2 wavefronts per SIMD"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_lookup2.so timeout 300 $B 2>/dev/null | line
  echo "4 wavefronts per SIMD"; SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_lookup4.so timeout 300 $B 2>/dev/null | line
done
} | tee $O/c5_lookup_waves.txt
