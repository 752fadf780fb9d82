"""FP64 matrix-pipe issue-rate probes: accumulators in VGPRs vs AccVGPRs, 16x16x4 vs 4x4x4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safe_learning_amd import _hip
ctx = _hip.Context()
for per_cu in ("1", "2", "4"):
    os.environ["SL_PROBE_BLOCKS_PER_CU"] = per_cu
    for which, name in ((0, "mfma16_vgpr_acc"), (3, "mfma16_agpr_acc"), (4, "mfma4x4x4"), (5, "mfma4x4x4_distinct_operands"), (6, "mfma4x4x4_64acc"), (7, "mfma4x4x4_plus_2_valu_fma_each"), (8, "mfma4x4x4_lds_fed")):
        print(per_cu, name, ctx.debug_fp64_rate(which, 20000), flush=True)
