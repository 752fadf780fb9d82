"""The build-time code audit of the inline-asm MFMA kernels (tools/audit_gp4.py::audit_in_place) on
small hand-written listings: every rule has to accept the pattern the kernels use and reject the
pattern it exists for.  (The shipped listing itself is audited by every build, safe_learning_amd/_build.py.)"""

import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import audit_gp4                                                      # noqa: E402

NAME = "_Z11k_bellman4sILi4ELi2ELb1EEvTEST"
MFMA = "\tv_mfma_f64_4x4x4_4b_f64 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]"


def listing(loop_lines, tmp_path):
    text = "\n".join(["\t.text", NAME + ":", "\ts_load_dwordx2 s[0:1], s[4:5], 0x0", ".LBB0_1:"]
                     + loop_lines + ["\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm", ""])
    path = tmp_path / "listing.s"
    path.write_text(text)
    return str(path)


def mfma(acc, a=2, b=4):
    return MFMA % (acc, acc + 1, a, a + 1, b, b + 1, acc, acc + 1)


CLEAN = ["\tds_read_b128 v[4:7], v60", "\ts_waitcnt lgkmcnt(0)", mfma(100), mfma(102), mfma(100, b=6),
         "\tbuffer_load_dwordx4 v[2:5], v61, s[8:11], 0 offen", "\ts_barrier"]


def problems(lines, tmp_path):
    return audit_gp4.audit_in_place(listing(lines, tmp_path))[1]


def test_clean_loop_passes(tmp_path):
    report, found = audit_gp4.audit_in_place(listing(CLEAN, tmp_path))
    assert not found and len(report) == 1 and "3 MFMAs on 4 accumulator registers" in report[0]


def test_no_kernel_is_a_problem(tmp_path):
    path = tmp_path / "empty.s"
    path.write_text("\t.text\n")
    assert audit_gp4.audit_in_place(str(path))[1]


@pytest.mark.parametrize("bad, what", [
    ("\tv_mov_b64_e32 v[100:101], 0", "accumulators touched"),
    ("\tv_add_f64 v[8:9], v[102:103], v[10:11]", "accumulators touched"),
    ("\tscratch_store_dwordx2 off, v[20:21], off", "scratch traffic"),
    ("\tv_mfma_f64_4x4x4_4b_f64 v[104:105], v[2:3], v[4:5], v[100:101]", "out-of-place"),
])
def test_rejects(bad, what, tmp_path):
    found = problems(CLEAN[:3] + [bad] + CLEAN[3:], tmp_path)
    assert any(what in p for p in found), found


def test_wait_states_in_front_of_an_mfma_source(tmp_path):
    scale = "\tv_mul_f64 v[2:3], v[2:3], v[30:31]"
    # the multiply right in front of the MFMA that reads v[2:3], or one slot away: rejected
    for between in ([], ["\ts_nop 0"], ["\tds_read_b64 v[40:41], v60"]):
        found = problems(CLEAN[:2] + [scale] + between + CLEAN[2:], tmp_path)
        assert any("MFMA source written right in front" in p for p in found), (between, found)
    # two issue slots (s_nop 1 counts two) are what the kernels leave
    for between in (["\ts_nop 1"], ["\ts_nop 0", "\ts_nop 0"], ["\tds_read_b64 v[40:41], v60", "\ts_nop 0"]):
        assert not problems(CLEAN[:2] + [scale] + between + CLEAN[2:], tmp_path), between
    # a multiply into a register no MFMA of the next two slots reads is fine
    assert not problems(CLEAN[:2] + ["\tv_mul_f64 v[32:33], v[32:33], v[30:31]"] + CLEAN[2:], tmp_path)


def dma_loop(wait, loads=2, order="dma_first"):
    dma = ["\ts_mov_b32 m0, s20", "\ts_nop 0", "\tglobal_load_lds_dwordx4 v50, s[6:7]"]
    work = ["\tds_read_b128 v[4:7], v60", "\ts_waitcnt lgkmcnt(0)", mfma(100), mfma(102)]
    work += ["\tbuffer_load_dwordx4 v[2:5], v61, s[8:11], 0 offen"] * loads
    tail = ([wait] if wait else []) + ["\ts_add_i32 s47, s47, 2", "\ts_barrier"]
    if order == "dma_first":
        return dma + work + tail
    # rotated layout: the block with the barrier first, the copy is issued in the block behind it
    return work + tail + ["\ts_cbranch_scc1 .LBB0_9"] + dma


def test_direct_to_lds_copies_need_a_counted_wait(tmp_path):
    for order in ("dma_first", "rotated"):
        assert not problems(dma_loop("\ts_waitcnt vmcnt(2)", order=order), tmp_path), order
        assert not problems(dma_loop("\ts_waitcnt vmcnt(0)", order=order), tmp_path), order
        found = problems(dma_loop("\ts_waitcnt vmcnt(3)", order=order), tmp_path)
        assert any("only 2 loads behind" in p for p in found), (order, found)
        found = problems(dma_loop(None, order=order), tmp_path)
        assert any("no s_waitcnt vmcnt(N) in front of the barrier" in p for p in found), (order, found)


def test_shipped_listing_passes():
    path = os.path.join(ROOT, "safe_learning_amd", "build", "sl_bellman4",
                        "sl_bellman4-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(path):
        pytest.skip("no build directory (the library was built elsewhere)")
    report, found = audit_gp4.audit_in_place(path)
    assert not found and len(report) >= 20


# ---- audit(): k_gp_sweep4 owns a[0:255]; only its asm groups may name them -------------------
GP_NAME = "_Z11k_gp_sweep4ILi4ELi1ELb1EEvTEST"


def gp_listing(streams, extra, tmp_path):
    lines = ["\t.text", GP_NAME + ":"]
    for k, stream in enumerate(streams):
        lines += [".LBB0_%d:" % (k + 1)] + stream + ["\ts_cbranch_scc1 .LBB0_%d" % (k + 1)]
    lines += extra + ["\ts_endpgm", ""]
    path = tmp_path / "gp.s"
    path.write_text("\n".join(lines))
    return str(path)


def gp_stream(n=64, inside=()):
    group = ["\t;;#ASMSTART"] + ["\tv_mfma_f64_4x4x4_4b_f64 a[0:1], v[2:3], v[4:5], a[0:1]"] * 8 + ["\t;;#ASMEND"]
    return ["\tds_read_b128 v[4:7], v60"] + list(inside) + group * (n // 8)


def test_gp4_audit_accepts_the_asm_groups(tmp_path):
    report, found = audit_gp4.audit(gp_listing([gp_stream()] * 2, [], tmp_path), min_loops=2)
    assert not found and "128 MFMAs" in report[0] and "in 2 MFMA loops" in report[0]


@pytest.mark.parametrize("streams, extra, what", [
    ([gp_stream()] * 2, ["\tv_accvgpr_read_b32 v9, a17"], "touches accumulator registers"),       # compiler AGPR use
    ([gp_stream()] * 2, ["\tv_mfma_f64_4x4x4_4b_f64 a[8:9], v[2:3], v[4:5], a[8:9]"], "touches accumulator"),
    ([gp_stream(inside=["\tscratch_load_dword v1, off, off"])] * 2, [], "scratch instructions inside the MFMA loops"),
    ([gp_stream(inside=["\tv_readlane_b32 s4, v200, 3"] * 17)] * 2, [], "v_readlane / v_writelane in one MFMA stream"),
    ([gp_stream()], [], "only 1 MFMA loops recognised"),
    ([["\ts_nop 0"]] * 2, [], "no MFMA found"),
])
def test_gp4_audit_rejects(streams, extra, what, tmp_path):
    found = audit_gp4.audit(gp_listing(streams, extra, tmp_path), min_loops=2)[1]
    assert any(what in p for p in found), found


def test_gp4_audit_allows_scratch_and_lane_reads_outside_the_streams(tmp_path):
    extra = ["\tscratch_load_dword v1, off, off", "\tv_readlane_b32 s4, v200, 3"] * 40
    assert not audit_gp4.audit(gp_listing([gp_stream()] * 2, extra, tmp_path), min_loops=2)[1]


def test_gp4_audit_rejects_a_copy_into_an_mfma_source(tmp_path):
    """Round 5: where the ways out of the diagonal stream joined, the compiler copied the k_x
    fragments into other registers right in front of the first MFMA group of the blocks below -
    the inline-asm MFMA read the registers' old contents (wrong |a|^2 on the GPU).  A vector-ALU
    write of an A / B source fewer than two issue slots ahead of the MFMA is refused."""
    copy = "\tv_mov_b64_e32 v[4:5], v[20:21]"
    for between in ([], ["\ts_waitcnt lgkmcnt(0)"], ["\ts_nop 0"]):
        found = audit_gp4.audit(gp_listing([gp_stream(inside=[copy] + between)] * 2, [], tmp_path), min_loops=2)[1]
        assert any("MFMA source written right in front" in p for p in found), (between, found)
    for between in (["\ts_nop 1"], ["\tds_read_b128 v[40:43], v60", "\ts_waitcnt lgkmcnt(0)"]):
        assert not audit_gp4.audit(gp_listing([gp_stream(inside=[copy] + between)] * 2, [], tmp_path),
                                   min_loops=2)[1], between
    # a copy into a register that the next MFMAs do not read is none of the audit's business
    other = "\tv_mov_b64_e32 v[40:41], v[20:21]"
    assert not audit_gp4.audit(gp_listing([gp_stream(inside=[other])] * 2, [], tmp_path), min_loops=2)[1]
