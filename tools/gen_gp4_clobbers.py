"""Regenerates safe_learning_amd/csrc/sl_gp4_clobbers.h (register-name literals for the clobber
lists of k_gp_sweep4's inline-asm MFMA groups)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
lines = ['// sl_gp4_clobbers.h - accumulator registers of k_gp_sweep4 by (row block r, cell-block pair h):',
         '// acc(r, cb, rot) = a[2 i : 2 i + 1], i = (r * 4 + cb) * 4 + rot, so cell blocks 2h, 2h + 1 of row block r',
         '// live in a[32 r + 16 h : 32 r + 16 h + 15].  Clobber lists of inline asm must be string literals,',
         '// hence this table (generated: tools/gen_gp4_clobbers.py).',
         '#pragma once']
for r in range(8):
    for h in range(2):
        base = 32 * r + 16 * h
        lines.append('#define SL_GP4_CL_%d_%d %s' % (r, h, ", ".join('"a%d"' % (base + k) for k in range(16))))
    lines.append('#define SL_GP4_CL_%d SL_GP4_CL_%d_0, SL_GP4_CL_%d_1' % (r, r, r))
with open(os.path.join(HERE, "..", "safe_learning_amd", "csrc", "sl_gp4_clobbers.h"), "w") as f:
    f.write("\n".join(lines) + "\n")
