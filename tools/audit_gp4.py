"""Audit of sl_gp4.hip's compiled code (run by safe_learning_amd._build after every build).

k_gp_sweep4 keeps its 128 FP64 accumulators at fixed accumulator registers a[0:255] that only
the inline-asm MFMA groups may touch (k_bellman4 of sl_bellman4.hip: 48 at a[0:95], same rules).  The compiler must therefore never use an accumulator
register on its own (VGPR spills to AGPRs are switched off with -amdgpu-spill-vgpr-to-agpr=0, this
script proves it), and the MFMA streams must be free of scratch traffic (and almost free of lane-spill reads).  k_bellman4
owns a[0:95] only: the compiler may use the accumulator registers above.
Usage: python tools/audit_gp4.py <file.s> [kernel prefix [min loops [owned registers]]]   (exit code 1 on a violation)."""
import re
import sys


def _touches_owned(code, owned):
    """Does the instruction name an accumulator register below `owned`?"""
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]|(?<![\w.])a(\d+)\b", code):
        first = int(m.group(1) if m.group(1) is not None else m.group(3))
        if first < owned:
            return True
    return False


def _vregs(text):
    out = set()
    for mm in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if mm.group(3) is not None:
            out.add(int(mm.group(3)))
        else:
            out.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
    return out


def mfma_sources_written_too_late(body, slots_needed=2):
    """The MFMAs are inline asm: the compiler's hazard recogniser does not look inside, and the
    hardware does not interlock a vector-ALU write of an A / B source register with an FP64 MFMA
    that reads it fewer than ``slots_needed`` issue slots later (round 5: register copies that the
    compiler placed where control flow joined in front of an MFMA group - the MFMA read the old
    register contents).  Returns the offending (producer, MFMA) pairs of a kernel body."""
    code = [l.split(";")[0].strip() for l in body if not l.strip().startswith(";;")]
    code = [c for c in code if c and not c.startswith(".") and not c.endswith(":")]
    close = []
    for i, c in enumerate(code):
        if not c.startswith("v_mfma_f64_4x4x4"):
            continue
        ops = c.split(None, 1)[1].split(", ")
        sources = _vregs(ops[1]) | _vregs(ops[2])
        slots = 0
        for back in range(1, slots_needed + 2):
            if i - back < 0 or slots >= slots_needed:
                break
            prev = code[i - back]
            if prev.startswith("s_nop"):
                slots += int(prev.split()[1]) + 1
                continue
            if prev.startswith("v_") and not prev.startswith("v_mfma") and " " in prev:
                if _vregs(prev.split(None, 1)[1].split(", ")[0]) & sources:
                    close.append("%s -> %s" % (prev, c))
            slots += 1
    return close


def audit(path, prefix="_Z11k_gp_sweep4", min_loops=8, owned=256):
    with open(path) as f:
        text = f.read()
    kernels = re.split(r"\n(?=" + prefix + ")", text)
    problems, report = [], []
    for chunk in kernels:
        m = re.match(r"(" + prefix + r"\w+):", chunk)
        if not m:
            continue
        name = m.group(1)
        body = chunk.split("s_endpgm")[0].split("\n")
        in_asm, mfma, outside = False, 0, []
        # Kernels that own only part of the file (owned < 256) read their accumulators out after
        # the MFMA loop: the registers must be left alone from the first zeroing asm block to the
        # last read-out block; afterwards (epilogue) the compiler may use them.
        first_zero = last_read = None
        if owned < 256:
            inside = False
            for i, line in enumerate(body):
                if ";;#ASMSTART" in line:
                    inside = True
                elif ";;#ASMEND" in line:
                    inside = False
                elif inside and first_zero is None and re.search(r"v_accvgpr_write_b32 a\d+, 0", line):
                    first_zero = i
                elif inside and "v_accvgpr_read_b32" in line:
                    last_read = i
        for i, line in enumerate(body):
            code = line.split(";")[0] if not line.strip().startswith(";;") else line
            guarded = owned == 256 or (first_zero is not None and last_read is not None
                                       and first_zero <= i <= last_read)
            if ";;#ASMSTART" in line:
                in_asm = True
            elif ";;#ASMEND" in line:
                in_asm = False
            elif not in_asm and guarded and _touches_owned(code, owned):
                outside.append(line.strip())
            if "v_mfma_f64_4x4x4" in line:
                mfma += 1
                if not in_asm:
                    outside.append(line.strip())
        scratch = sum("scratch_" in l for l in body)
        # The MFMA streams: straight-line code (no label, no branch) with at least 64 MFMAs - a
        # slab-pair loop body or, since the chunks are fully unrolled, a whole chunk.  They must be
        # free of scratch traffic; scalar values spilled to VGPR lanes may be read back (the byte
        # offsets of the row blocks), but only a handful per stream.
        hot, seg = [], []
        for line in body:
            if re.match(r"\.LBB\d+_\d+:", line) or re.search(r"\bs_c?branch", line):
                if sum("v_mfma" in l for l in seg) >= 64:
                    hot.append(seg)
                seg = []
            else:
                seg.append(line)
        if sum("v_mfma" in l for l in seg) >= 64:
            hot.append(seg)
        lane_worst = max([sum(("v_readlane" in l or "v_writelane" in l) for l in seg) for seg in hot] or [0])
        lane_ops = sum(sum(("v_readlane" in l or "v_writelane" in l) for l in seg) for seg in hot)
        hot_scratch = sum(sum("scratch_" in l for l in seg) for seg in hot)
        report.append("%s: %d MFMAs, %d scratch ops (%d in MFMA loops), %d AGPR references outside "
                      "the asm groups, %d lane-spill ops in %d MFMA loops"
                      % (name, mfma, scratch, hot_scratch, len(outside), lane_ops, len(hot)))
        if outside:
            problems.append("%s: the compiler touches accumulator registers: %s" % (name, outside[:3]))
        if hot_scratch:
            problems.append("%s: %d scratch instructions inside the MFMA loops" % (name, hot_scratch))
        if len(hot) < min_loops:
            problems.append("%s: only %d MFMA loops recognised (expected at least %d)"
                            % (name, len(hot), min_loops))
        if lane_worst > 16:
            problems.append("%s: %d v_readlane / v_writelane in one MFMA stream" % (name, lane_worst))
        if mfma == 0:
            problems.append("%s: no MFMA found" % name)
        late = mfma_sources_written_too_late(body)
        if late:
            problems.append("%s: MFMA source written right in front of it: %s" % (name, late[:2]))
    if not report:
        problems.append("no %s kernel found in %s" % (prefix, path))
    return report, problems


def audit_in_place(path, prefix=r"_Z1[017]k_bellman4(?:s|_policy)?I"):
    """k_bellman4 keeps its accumulators as ordinary register values that inline-asm MFMAs update
    in place.  The compiler sees those asm statements as opaque: a register copy or a spill of an
    accumulator inside the MFMA loop would read a result the hardware has not retired yet (no
    interlock for the asm's outputs).  Check, per instantiation, that the innermost loop with MFMAs
    holds nothing but in-place MFMAs, LDS fragment reads, buffer loads and scalar instructions."""
    with open(path) as f:
        text = f.read()
    report, problems = [], []
    for chunk in re.split(r"\n(?=" + prefix + ")", text):
        m = re.match(r"(" + prefix + r"\w+):", chunk)
        if not m:
            continue
        name = m.group(1)
        body = chunk.split("s_endpgm")[0].split("\n")
        labels = {}
        for i, line in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", line)
            if mm:
                labels[mm.group(1)] = i
        loops = []
        for i, line in enumerate(body):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", line)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                start = labels[mm.group(1)]
                if any("v_mfma" in l for l in body[start:i + 1]):
                    loops.append((start, i))
        # innermost MFMA loops: those that contain no other MFMA loop (k_bellman4 has one chunk
        # loop, k_bellman4_policy one per number of action slots)
        inner = [(a, b) for (a, b) in loops
                 if not any((c, e) != (a, b) and a <= c and e <= b for (c, e) in loops)]
        if not inner:
            problems.append("%s: no MFMA loop found" % name)
            continue
        for start, end in inner:
            best = body[start:end + 1]
            mfmas = [l for l in best if "v_mfma" in l]
            moved = [l.strip() for l in mfmas if not re.search(
                r"v_mfma_f64_4x4x4_4b_f64 (v\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], \1", l)]
            # registers the MFMAs accumulate into, and every other instruction of the loop naming one
            accs = set()
            for l in mfmas:
                mm = re.search(r"v_mfma_f64_4x4x4_4b_f64 v\[(\d+):(\d+)\]", l)
                accs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            touching = []
            for l in best:
                code = l.split(";")[0]
                if "v_mfma" in code or not code.strip() or code.strip().startswith("."):
                    continue
                regs = set()
                for mm in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", code):
                    if mm.group(3) is not None:
                        regs.add(int(mm.group(3)))
                    else:
                        regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                if regs & accs:
                    touching.append(code.strip())
            spills = [l.strip() for l in best if "scratch_" in l]
            # The MFMAs are inline asm: the compiler's hazard recogniser does not see them.  A
            # vector ALU instruction that writes an A / B source register must not sit in the two
            # issue slots in front of the MFMA that reads it.
            def regs_of(text):
                out = set()
                for mm in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
                    if mm.group(3) is not None:
                        out.add(int(mm.group(3)))
                    else:
                        out.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                return out
            code_lines = [l.split(";")[0].strip() for l in best]
            code_lines = [c for c in code_lines if c and not c.startswith(".") and not c.endswith(":")]
            close = []
            for i, c in enumerate(code_lines):
                if not c.startswith("v_mfma"):
                    continue
                ops = c.split(None, 1)[1].split(", ")
                sources = regs_of(ops[1]) | regs_of(ops[2])
                slots = 0                        # issue slots between the producer and the MFMA
                for back in range(1, 4):
                    if i - back < 0 or slots >= 2:
                        break
                    prev = code_lines[i - back]
                    if prev.startswith("s_nop"):
                        slots += int(prev.split()[1]) + 1
                        continue
                    if prev.startswith("v_") and not prev.startswith("v_mfma") and " " in prev:
                        if regs_of(prev.split(None, 1)[1].split(", ")[0]) & sources:
                            close.append("%s -> %s" % (prev, c))
                    slots += 1
            # Direct global -> LDS loads (k_bellman4s' chunk copy) are inline asm too: the
            # compiler neither counts them nor waits for them.  In front of the loop's barrier there
            # has to be an s_waitcnt vmcnt(N) with N <= the number of loads issued behind the last
            # such copy (walking backwards from the barrier, around the loop's layout).
            dma_problem = None
            if any("global_load_lds" in c for c in code_lines):
                barriers = [i for i, c in enumerate(code_lines) if c.startswith("s_barrier")]
                if not barriers:
                    dma_problem = "direct-to-LDS loads in a loop without a barrier"
                for b in barriers:
                    wait = None
                    for back in range(1, 10):
                        mm = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", code_lines[b - back]) if b - back >= 0 else None
                        if mm:
                            wait, wait_at = int(mm.group(1)), b - back
                            break
                        if b - back >= 0 and code_lines[b - back].startswith(("v_mfma", "buffer_", "global_", "ds_")):
                            break
                    if wait is None:
                        dma_problem = "no s_waitcnt vmcnt(N) in front of the barrier of a loop with direct-to-LDS loads"
                        break
                    order = list(range(wait_at - 1, -1, -1)) + list(range(len(code_lines) - 1, b, -1))
                    younger = 0
                    for i in order:
                        c = code_lines[i]
                        if "global_load_lds" in c:
                            break
                        if c.startswith(("buffer_load", "global_load")):
                            younger += 1
                    if wait > younger:
                        dma_problem = ("s_waitcnt vmcnt(%d) in front of the barrier, but only %d loads behind "
                                       "the last direct-to-LDS copy" % (wait, younger))
                        break
            if dma_problem:
                problems.append("%s: %s" % (name, dma_problem))
            report.append("%s: %d MFMAs on %d accumulator registers in the chunk loop, %d out of "
                          "place, %d other instructions touch an accumulator, %d scratch accesses, "
                          "%d sources written within two slots of their MFMA"
                          % (name, len(mfmas), len(accs), len(moved), len(touching), len(spills),
                             len(close)))
            if close:
                problems.append("%s: MFMA source written right in front of it: %s" % (name, close[:2]))
            if moved:
                problems.append("%s: out-of-place MFMAs: %s" % (name, moved[:2]))
            if touching:
                problems.append("%s: accumulators touched inside the MFMA loop: %s" % (name, touching[:3]))
            if spills:
                problems.append("%s: scratch traffic inside the MFMA loop: %s" % (name, spills[:2]))
    if not report:
        problems.append("no %s kernel found in %s" % (prefix, path))
    return report, problems


if __name__ == "__main__":
    report, problems = audit(sys.argv[1], *(sys.argv[2:3] or ["_Z11k_gp_sweep4"]),
                             *[int(v) for v in sys.argv[3:5]])
    print("\n".join(report))
    if problems:
        print("AUDIT FAILED:\n" + "\n".join(problems))
        sys.exit(1)
    print("audit ok")
