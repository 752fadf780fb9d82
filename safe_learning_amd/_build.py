"""Build libslhip.so (HIP/gfx950) in-tree with hipcc.

    python -m safe_learning_amd._build [--verbose]

The library is built next to this file so that it travels to the GPU box with the source
tree.  ``-ffp-contract=off`` is part of the numerical contract: the per-cell arithmetic must
round exactly like the float64 oracle (one rounding per multiply and per add).
"""

import glob
import os
import subprocess
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# translation units: (object stem, source, extra flags).  sl_gp4.hip is compiled once per state
# dimension (its unrolled MFMA streams make one instantiation a minute of compile time; the four
# jobs run side by side).
GP4_R = 4                                        # row blocks per wavefront of k_gp_sweep4 (sl_gp4.hip)
GP4_FLAGS = ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-save-temps=obj", "-DSL_GP4_R=%d" % GP4_R]
UNITS = [("sl_kernels", "sl_kernels.hip", []), ("sl_gp", "sl_gp.hip", []),
         ("sl_bellman", "sl_bellman.hip", []), ("sl_bellman4", "sl_bellman4.hip", ["-save-temps=obj"]),
         ("sl_nn", "sl_nn.hip", []), ("sl_comm", "sl_comm.hip", []),
         ("sl_gp_small", "sl_gp_small.hip", []), ("sl_det_rows", "sl_det_rows.hip", []),
         ("sl_level", "sl_level.hip", []), ("sl_adaptive", "sl_adaptive.hip", []),
         ("sl_sample", "sl_sample.hip", []), ("sl_region", "sl_region.hip", []),
         ("sl_succ", "sl_succ.hip", []), ("sl_policy_net", "sl_policy_net.hip", [])]
UNITS += [("sl_gp4_d%d" % dim, "sl_gp4.hip", GP4_FLAGS + ["-DSL_GP4_DIM=%d" % dim]) for dim in (1, 2, 3, 4)]
LIB = os.path.join(HERE, "libslhip.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _listing(objdir, stem):
    found = glob.glob(os.path.join(objdir, stem, "*-hip-amdgcn*gfx950*.s"))
    if not found:
        raise RuntimeError("device assembly of %s not found in %s" % (stem, objdir))
    return found[0]


def _audit_gp4(objdir, verbose):
    """k_gp_sweep4 owns the accumulator registers through inline asm: prove on the generated code
    that the compiler never uses one and keeps the MFMA streams free of spill traffic
    (tools/audit_gp4.py::audit), for every dimension's translation unit."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_gp4
    for dim in (1, 2, 3, 4):
        # (one MFMA stream per number of active row blocks; 32 accumulator registers per row block)
        report, problems = audit_gp4.audit(_listing(objdir, "sl_gp4_d%d" % dim), "_Z11k_gp_sweep4", GP4_R,
                                           32 * GP4_R)
        if verbose:
            print("\n".join(report))
        if problems:
            raise RuntimeError("sl_gp4.hip (d = %d) failed its code audit:\n" % dim + "\n".join(problems))


def _audit_bellman4(objdir, verbose):
    """k_bellman4: accumulators are register values updated by opaque inline-asm MFMAs - see
    tools/audit_gp4.py::audit_in_place."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_gp4
    report, problems = audit_gp4.audit_in_place(_listing(objdir, "sl_bellman4"))
    if verbose:
        print("\n".join(report))
    if problems:
        raise RuntimeError("sl_bellman4.hip failed its code audit:\n" + "\n".join(problems))


def build(verbose=False, force=False, run_audits=True, lib=None, only=None):
    """``run_audits=False`` / ``lib=...`` / ``only=[stems]``: development builds only (instrumented
    kernels whose listing the audits do not describe, written next to the shipped library; the
    units not named in ``only`` are taken from the shipped build) - see tools/build_variant.sh."""
    LIB = lib or globals()["LIB"]
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(ROOT, "include", "sl_hip.h"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the flag set is part of what the library is: a change of SL_GP4_R / SL_EXTRA_FLAGS / HIPCC
    # rebuilds like a source edit does (the stamp travels with the library)
    stamp_path = LIB + ".flags"
    stamp = "GP4_R=%d\nEXTRA=%s\nHIPCC=%s\n" % (GP4_R, os.environ.get("SL_EXTRA_FLAGS", ""), hipcc)
    stamp_ok = os.path.exists(stamp_path) and open(stamp_path).read() == stamp
    if not force and stamp_ok and not _newer(LIB, deps):
        return LIB
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
             "-fvisibility=hidden",           # exports = the SL_API declarations of include/sl_hip.h
             "-I" + os.path.join(ROOT, "include"), "-I" + csrc]
    flags += os.environ.get("SL_EXTRA_FLAGS", "").split()         # experiments: -DSL_GP_CFG... etc.
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    # one hipcc per translation unit, all at once, then one link.  Each unit has its own directory
    # (-save-temps=obj names the listing after the SOURCE file).
    objdir = os.path.join(HERE, "build" if lib is None else "build_dev")

    def compile_cmd(stem, src, extra):
        os.makedirs(os.path.join(objdir, stem), exist_ok=True)
        return [hipcc] + flags + extra + ["-c", os.path.join(csrc, src), "-o",
                                          os.path.join(objdir, stem, stem + ".o")]

    # a unit is recompiled when its source, a header or the flag set changed (force: all of them)
    headers = [f for f in deps if f.endswith(".h")]
    jobs = []
    for stem, src, extra in UNITS:
        if only is not None and stem not in only:
            continue
        obj = os.path.join(objdir, stem, stem + ".o")
        if not force and stamp_ok and not _newer(obj, headers + [os.path.join(csrc, src)]):
            continue
        cmd = compile_cmd(stem, src, extra)
        if verbose:
            print(" ".join(cmd))
        jobs.append((stem, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for stem, proc in jobs:
        out, _ = proc.communicate()
        if verbose or proc.returncode != 0:
            sys.stderr.write(out)
        failed = failed or proc.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objects = {stem: os.path.join(objdir if only is None or stem in only else os.path.join(HERE, "build"),
                                  stem, stem + ".o") for stem, _, _ in UNITS}
    # The 4x4x4 kernels rely on properties of the generated code that only the audits can prove
    # (inline asm owns accumulator registers).  A toolchain that schedules or names things
    # differently must not make the package unbuildable: the kernel in question is compiled out
    # (-DSL_NO_GP4 / -DSL_NO_BELLMAN4: the 16x16x4 kernels take over) and the build warns.
    for audit, src, macro, stems in ((_audit_gp4, "sl_gp4.hip", "SL_NO_GP4",
                                      ["sl_gp4_d%d" % dim for dim in (1, 2, 3, 4)]),
                                     (_audit_bellman4, "sl_bellman4.hip", "SL_NO_BELLMAN4", ["sl_bellman4"])):
        if not run_audits:
            continue
        if os.environ.get("SL_FORCE_AUDIT_FAILURE") == macro:       # exercised by the tests
            problem = "forced by SL_FORCE_AUDIT_FAILURE"
        else:
            try:
                audit(objdir, verbose)
                continue
            except RuntimeError as exc:
                problem = str(exc)
        warnings.warn("%s failed its code audit; building without it (-D%s): %s"
                      % (src, macro, problem), RuntimeWarning)
        for stem in stems:
            objects.pop(stem)
        stem = macro.lower()
        res = subprocess.run(compile_cmd(stem, src, ["-D" + macro]), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout)
            raise RuntimeError("hipcc failed on the fallback build of %s" % src)
        objects[stem] = os.path.join(objdir, stem, stem + ".o")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + sorted(objects.values()) + ["-ldl"]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed (exit %d)" % res.returncode)
    if only is None:
        with open(stamp_path, "w") as handle:
            handle.write(stamp)
    return LIB


if __name__ == "__main__":
    build(verbose="--verbose" in sys.argv, force=True)
    print(LIB)
