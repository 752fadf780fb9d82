"""Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 (operands and result), with and without the
A-block broadcast."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safe_learning_amd import _hip
ctx = _hip.Context()
eye = np.eye(64)
for mode in (0, 1, 2, 4):
    # wave p: a = one-hot at lane p, b_l = l + 1  -> d_r = (q + 1) for the B lane q paired with A lane p
    d = ctx.debug_mfma4(eye, np.tile(np.arange(1.0, 65.0), (64, 1)), np.zeros((64, 64)), mode)
    print("mode", mode)
    for p in range(64):
        r = np.flatnonzero(d[p])
        print("  A lane %2d -> D lanes %s  with B lanes %s" % (p, r.tolist(), (d[p][r] - 1).astype(int).tolist()))
