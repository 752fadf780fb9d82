#!/usr/bin/env python3
"""Diagnostic builds of k_gp_sweep4 (results meaningless): copies of sl_gp4.hip with one phase of the
generation / posterior-mean pass removed, built by tools/build_variant.sh into libslhip_<name>.so.
Timed with SL_GP4_SKIP=8 (no MFMA streams) the phases run alone; profiles/r04_summary.md section 2."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "safe_learning_amd", "csrc", "sl_gp4.hip")
OUT = os.path.join(ROOT, "safe_learning_amd", "build", "variants")

def sub(text, old, new):
    assert old in text, old
    return text.replace(old, new)

def variant(name):
    s = open(SRC).read()
    if name == "g_noload":       # generation: no global loads (inputs made up from the lane)
        s = sub(s, "if (q < p) xv[q] = XSG ? xs_glob[q * n_pad + j] : xs_l[q * n_pad + j];",
                "if (q < p) xv[q] = 1e-3 * (lane + q);")
        s = sub(s, "if (seed_w && ch < first_new) {", "if (false) {")
    elif name == "g_nostore":    # generation: one LDS store instead of sixteen
        s = sub(s, "(c < 12 ? w_lo : w_hi)[2 * c] = e;", "if (c == 15) w_lo[0] = e;")
    elif name == "g_noexp":      # generation: no exponentials
        s = sub(s, "exp_pair(-0.5 * z, fmin(bj - 0.5 * a2, 700.0), e, rho);", "e = z; rho = bj;")
    elif name == "g_norec":      # generation: no recurrence
        s = sub(s, "                        e *= rho;\n                        rho *= qstep;\n                    }\n                } else if (!direct)",
                "                    }\n                } else if (!direct)")
    elif name == "m_noload":     # mean pass: alpha' made up
        s = sub(s, "a0[s2] = ap[(8 * s2) * stride];", "a0[s2] = 1e-3 * (lane + s2);")
        s = sub(s, "a1[s2] = ap[(8 * s2 + 4) * stride];", "a1[s2] = 2e-3 * (lane + s2);")
    elif name == "m_nonop":      # mean pass: no trailing wait states
        s = sub(s, 'asm volatile("s_nop 15\\n\\ts_nop 7" : "+v"(macc[0])', 'asm volatile("" : "+v"(macc[0])')
    elif name == "nobarrier":    # chunk loop without its barrier
        s = sub(s, "generate(ch + 1, buf ^ 1, first_new_chunk, keep);\n                    __syncthreads();",
                "generate(ch + 1, buf ^ 1, first_new_chunk, keep);")
    elif name == "base":
        pass
    else:
        raise SystemExit("unknown variant " + name)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "sl_gp4_%s.hip" % name)
    open(path, "w").write(s)
    subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), name, "sl_gp4_d4", path])

if __name__ == "__main__":
    for n in sys.argv[1:]:
        variant(n)
