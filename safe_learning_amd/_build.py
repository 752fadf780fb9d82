"""Build libslhip.so (HIP/gfx950) in-tree with hipcc.

    python -m safe_learning_amd._build [--verbose]

The library is built next to this file so that it travels to the GPU box with the source
tree.  ``-ffp-contract=off`` is part of the numerical contract: the per-cell arithmetic must
round exactly like the float64 oracle (one rounding per multiply and per add).
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["sl_kernels.hip", "sl_gp.hip", "sl_bellman.hip", "sl_nn.hip"]
LIB = os.path.join(HERE, "libslhip.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False):
    csrc = os.path.join(HERE, "csrc")
    srcs = [os.path.join(csrc, s) for s in SOURCES if os.path.exists(os.path.join(csrc, s))]
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "sl_hip.h"))
    if not force and not _newer(LIB, deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
             "-I" + os.path.join(ROOT, "include"), "-I" + csrc]
    flags += os.environ.get("SL_EXTRA_FLAGS", "").split()         # experiments: -DSL_GP_CFG... etc.
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    # one hipcc per translation unit, all at once (the kernels are heavily templated: ~2-4 min of
    # compile time in total), then one link
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for obj, proc in jobs:
        out, _ = proc.communicate()
        if verbose or proc.returncode != 0:
            sys.stderr.write(out)
        failed = failed or proc.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [obj for obj, _ in jobs]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed (exit %d)" % res.returncode)
    return LIB


if __name__ == "__main__":
    build(verbose="--verbose" in sys.argv, force=True)
    print(LIB)
