"""Audit of sl_gp4.hip's compiled code (run by safe_learning_amd._build after every build).

k_gp_sweep4 keeps its 128 FP64 accumulators at fixed accumulator registers a[0:255] that only
the inline-asm MFMA groups may touch.  The compiler must therefore never use an accumulator
register on its own (VGPR spills to AGPRs are switched off with -amdgpu-spill-vgpr-to-agpr=0, this
script proves it), and the MFMA loops must be free of scratch and lane-spill traffic.  Usage: python tools/audit_gp4.py <file.s>   (exit code 1 on a violation)."""
import re
import sys


def audit(path):
    text = open(path).read()
    kernels = re.split(r"\n(?=_Z11k_gp_sweep4)", text)
    problems, report = [], []
    for chunk in kernels:
        m = re.match(r"(_Z11k_gp_sweep4\w+):", chunk)
        if not m:
            continue
        name = m.group(1)
        body = chunk.split("s_endpgm")[0].split("\n")
        in_asm, mfma, outside = False, 0, []
        for line in body:
            code = line.split(";")[0] if not line.strip().startswith(";;") else line
            if ";;#ASMSTART" in line:
                in_asm = True
            elif ";;#ASMEND" in line:
                in_asm = False
            elif not in_asm and re.search(r"\bv_accvgpr|\ba\[\d+:\d+\]|[ ,]a\d+\b", code):
                outside.append(line.strip())
            if "v_mfma_f64_4x4x4" in line:
                mfma += 1
                if not in_asm:
                    outside.append(line.strip())
        scratch = sum("scratch_" in l for l in body)
        # lane-spill traffic inside the innermost (slab pair) loops: from an "Inner Loop Header"
        # to the backward branch that closes it, if MFMAs lie in between
        hot = []
        for i, line in enumerate(body):
            if "Inner Loop Header" in line:
                label = None
                for back in range(i, max(i - 12, 0), -1):
                    mm = re.match(r"(\.LBB\d+_\d+):", body[back])
                    if mm:
                        label = mm.group(1)
                        break
                seg = []
                for l in body[i:]:
                    seg.append(l)
                    if label and re.search(r"s_cbranch_\w+ " + re.escape(label) + r"\b", l):
                        break
                if any("v_mfma" in l for l in seg):
                    hot.append(seg)
        lane_ops = sum(sum(("v_readlane" in l or "v_writelane" in l) for l in seg) for seg in hot)
        hot_scratch = sum(sum("scratch_" in l for l in seg) for seg in hot)
        report.append("%s: %d MFMAs, %d scratch ops (%d in MFMA loops), %d AGPR references outside "
                      "the asm groups, %d lane-spill ops in %d MFMA loops"
                      % (name, mfma, scratch, hot_scratch, len(outside), lane_ops, len(hot)))
        if outside:
            problems.append("%s: the compiler touches accumulator registers: %s" % (name, outside[:3]))
        if hot_scratch:
            problems.append("%s: %d scratch instructions inside the MFMA loops" % (name, hot_scratch))
        if len(hot) < 8:
            problems.append("%s: only %d MFMA loops recognised (expected one per chunk variant)"
                            % (name, len(hot)))
        if lane_ops:
            problems.append("%s: %d v_readlane / v_writelane in the MFMA loops" % (name, lane_ops))
        if mfma == 0:
            problems.append("%s: no MFMA found" % name)
    if not report:
        problems.append("no k_gp_sweep4 kernel found in %s" % path)
    return report, problems


if __name__ == "__main__":
    report, problems = audit(sys.argv[1])
    print("\n".join(report))
    if problems:
        print("AUDIT FAILED:\n" + "\n".join(problems))
        sys.exit(1)
    print("audit ok")
