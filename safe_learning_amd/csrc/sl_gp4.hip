// sl_gp4.hip - the GP-dynamics Lyapunov sweep on v_mfma_f64_4x4x4_4b_f64 (training sets > 256
// points, closed-form policy and quadratic V: the headline path).  Same mathematics, data layout
// in HBM and epilogue as k_gp_sweep (sl_gp.hip); what differs is how the FP64 matrix pipe is fed:
//
//  * v_mfma_f64_16x16x4_f64 issues every ~85-100 cycles on gfx950 (2048 flops), the 4-block
//    v_mfma_f64_4x4x4_4b_f64 every 16.3 (512 flops): 47 against 76 TFLOP/s in isolation.  Its
//    four blocks are independent 4x4x4 products (block b = lanes 4b..4b+3 of every row of 16
//    lanes, tests/test_gpu_lyapunov.py::test_mfma_4x4x4_block_layout), so a 16-row x 16-cell x 4
//    tile takes four instructions: the A fragment (rows) stays, the k_x fragment (cells) is read
//    from LDS rotated by 0/4/8/12 lanes per row of 16.
//  * FP64 VALU work shares that pipe (measured: no overlap with FP64 MFMAs, own wave or partner).
//    The workgroup: four wavefronts with 256 registers each - 4 row blocks x 4 cell blocks x
//    4 rotations = 64 FP64 accumulators per lane in a[0:127] - 256-row panels and 79 KB of LDS,
//    so that TWO workgroups share a CU and cover each other's non-MFMA phases (generation,
//    barrier, per-tile prologue and epilogue); training inputs and alpha' come from L2.
//    (Round 2's shape - one wavefront per SIMD with the whole 512-register file, 512-row panels,
//    24 instead of 40 chunk generations per tile, its non-MFMA phases riding in the MFMA stream as
//    "fillers" - measured 2 % slower, profiles/r03_summary.md, and left the source in round 4.)
//  * The accumulators sit at FIXED accumulator registers and are only touched by inline-asm MFMA
//    groups.  With C++ accumulator variables every branch around the MFMAs (the lower triangle
//    makes the set of active row blocks chunk dependent) costs phi copies, out-of-place MFMAs and
//    spills (measured: 36 instead of 62 TFLOP/s for the GEMM phase).
//  * One straight-line body per number of active row blocks (template R0), the slab pairs of a
//    chunk unrolled, operands software pipelined: A fragments (buffer loads with scalar offsets)
//    two slab pairs ahead in three register sets, k_x fragments one rotation ahead - and the load
//    instructions placed BETWEEN the MFMAs of a group (rotation<>), not in front of it.
//  * k_x generation: where the GP input [x, policy(x)] / lengthscales is affine in the cell index
//    along the wavefront's 16 cells (a row of the last grid axis, policy linear or saturated) the
//    RBF values form a Gaussian sequence e_{c+1} = e_c rho_c, rho_{c+1} = rho_c Q: two
//    exponentials per (training point, 16 cells) instead of 16 (lane = training point).  The
//    cells are split into affine runs by a test on the values themselves (a kink of the
//    saturation or the end of a grid row restarts the recurrence); inputs that are not
//    piecewise affine (explicit points, table policies) take one exponential per (point, cell).
//    The posterior mean k_x . alpha' is a 4 x 16 x 64 product per chunk and wavefront: 16 more
//    MFMAs on the LDS copy of each new chunk.
//
// Measured in tools/gp_lab.hip (1 Mi cells, 1024 points): 16x16x4 structure 49.5 TFLOP/s, round 2's
// 4x4x4 kernel 60.5 (profiles/r02_summary.md); round 3: 62.4 TFLOP/s at 128^4 (profiles/r03_summary.md).
#include "sl_common.h"
#include "sl_gp4_clobbers.h"

#ifdef SL_NO_GP4
// Compiled out: the build's audit of the fixed-accumulator code failed on this toolchain
// (safe_learning_amd/_build.py); every GP sweep then runs k_gp_sweep (sl_gp.hip).
bool sl_gp4_supports(const SlDevModel&) { return false; }
int sl_gp4_sweep_launch(sl_ctx* ctx, const SlDevModel&, int64_t, int64_t, const uint64_t*, const double*,
                        uint64_t*, int*, double*, const double*) {
    return sl_fail(ctx, SL_ERR_UNSUPPORTED, "k_gp_sweep4 is compiled out of this build");
}
#else

typedef double sl_d2 __attribute__((ext_vector_type(2)));
typedef unsigned sl_u4 __attribute__((ext_vector_type(4)));

namespace gp4 {

// R row blocks per wavefront: 256 registers (64 accumulators, 256-row panels) and at most 80 KB of
// LDS per workgroup, so that TWO workgroups share a CU: they work on different tiles at their own
// pace, and whatever one of them does outside its MFMA stream (k_x generation, the barrier, the
// per-tile prologue and epilogue) runs under the other one's MFMAs.
#ifndef SL_GP4_R
#define SL_GP4_R 4
#endif
constexpr int R = SL_GP4_R, CB = 4, W = 4;
constexpr int NACC = R * CB * 4;           // FP64 accumulators per lane (2 registers each)
constexpr int C = 16 * CB;                 // cells per tile
constexpr int RP = 16 * R * W;             // rows per panel (512)
constexpr int RB = R * W;                  // row blocks per panel
// k_x chunk (64 training points x 64 cells) in LDS:
//   [slab pair 8][cell block CB][k 4][slot 16][slab of the pair 2]
// The slot of cell c16 in row k is (c16 + 4 (k >> 1)) & 15 and consecutive slab pairs are 4
// doubles apart modulo the banks.  ds_read_b128 serves the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md): rows 0/1 (and 2/3) must agree on the slot of a
// cell for the rotated fragment reads to be conflict free; the shift between the row pairs and
// the slab-pair skew leave the generation writes (lane = training point, 8 B, one cell per
// instruction) with 2-way conflicts only.
#ifndef SL_GP4_WRITE_SKEW
#define SL_GP4_WRITE_SKEW 1
#endif
constexpr int RUNS = 4;                    // affine runs per wavefront handled by the recurrence
constexpr int RUNC = SL_P + 2;             // per (wavefront, run): step[SL_P], a^2, Q
constexpr int KXS2 = CB * 128 + 4;
constexpr int KXBUF = 8 * KXS2;
static_assert(R == 4, "256-row panels (tools/audit_gp4.py is told the same number by the build)");

struct AFrag { sl_d2 v[R]; };
struct BFrag { sl_d2 v[CB]; };

// The 128 accumulators live at FIXED accumulator registers, acc(r, cb, rot) = a[2 i : 2 i + 1] with
// i = (r * CB + cb) * 4 + rot, and only the asm statements below touch them.  (As C++ values -
// builtin MFMAs or "+a" asm operands - the 128 live accumulators next to 128 operand registers
// make the compiler emit out-of-place MFMAs, accumulator copies at every branch and scratch
// spills inside the MFMA loops.)  Every statement lists the whole accumulator file as clobbered,
// so the compiler cannot keep a value of its own in an accumulator register across any of them;
// tools/audit_gp4.py (run by the build) proves on the generated code that it never uses one.
#define SL_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define SL_AGPRS_0_127                                                                             \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", SL_A10(1), SL_A10(2), SL_A10(3),   \
        SL_A10(4), SL_A10(5), SL_A10(6), SL_A10(7), SL_A10(8), SL_A10(9), SL_A10(10), SL_A10(11),   \
        "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define SL_ALL_AGPRS SL_AGPRS_0_127

template <int BASE>
__device__ __forceinline__ void acc_zero16() {
    asm volatile(
        "v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\t"
        "v_accvgpr_write_b32 a%c3, 0\n\tv_accvgpr_write_b32 a%c4, 0\n\tv_accvgpr_write_b32 a%c5, 0\n\t"
        "v_accvgpr_write_b32 a%c6, 0\n\tv_accvgpr_write_b32 a%c7, 0\n\tv_accvgpr_write_b32 a%c8, 0\n\t"
        "v_accvgpr_write_b32 a%c9, 0\n\tv_accvgpr_write_b32 a%c10, 0\n\tv_accvgpr_write_b32 a%c11, 0\n\t"
        "v_accvgpr_write_b32 a%c12, 0\n\tv_accvgpr_write_b32 a%c13, 0\n\tv_accvgpr_write_b32 a%c14, 0\n\t"
        "v_accvgpr_write_b32 a%c15, 0\n\ts_nop 3"
        :
        : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3), "i"(BASE + 4), "i"(BASE + 5),
          "i"(BASE + 6), "i"(BASE + 7), "i"(BASE + 8), "i"(BASE + 9), "i"(BASE + 10), "i"(BASE + 11),
          "i"(BASE + 12), "i"(BASE + 13), "i"(BASE + 14), "i"(BASE + 15)
        : SL_ALL_AGPRS);
}
template <int BASE = 0>
__device__ __forceinline__ void acc_zero_all() {
    acc_zero16<BASE>();
    if constexpr (BASE + 16 < 2 * NACC) acc_zero_all<BASE + 16>();
}
template <int N>
__device__ __forceinline__ double acc_read() {
    unsigned lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_accvgpr_read_b32 %1, a%c3\n\ts_nop 0"
                 : "=v"(lo), "=v"(hi)
                 : "i"(N), "i"(N + 1));
    return __hiloint2double((int)hi, (int)lo);
}
// |a|^2 of the panel's rows on the matrix pipe: with an accumulator register as BOTH operands of a 4x4x4 MFMA
// (lane (k, blk, low) holds a[row k of the register's four][cell low of its group]: the layout of
// the B operand and, transposed, of the A operand) the product is the 4 x 4 Gram matrix of the
// four rows, whose diagonal - lanes with k == low - is sum_k a[k][cell]^2.  Chained over the row
// blocks: gram[cb][rot] = sum_r acc(r, cb, rot)^T acc(r, cb, rot), sixteen independent chains
// (a dependent FP64 MFMA must not follow its producer directly), no accumulator reads, no folds
// across the lane groups.
template <int RI = 0, int J = 0>
__device__ __forceinline__ void acc_gram(double (&gram)[CB][4]) {
    constexpr int N = 2 * ((RI * CB + J / 4) * 4 + J % 4);
    if constexpr (RI == 0)
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, a[%c1:%c2], a[%c1:%c2], 0"
                     : "=v"(gram[J / 4][J % 4]) : "i"(N), "i"(N + 1));
    else
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, a[%c1:%c2], a[%c1:%c2], %0"
                     : "+v"(gram[J / 4][J % 4]) : "i"(N), "i"(N + 1));
    if constexpr (J + 1 < 4 * CB) acc_gram<RI, J + 1>(gram);
    else if constexpr (RI + 1 < R) acc_gram<RI + 1, 0>(gram);
}

// eight MFMAs of one (row block, rotation): both slabs of the pair for the four cell blocks.  The
// operands come straight from loads (the compiler's s_waitcnt precedes the statement); an
// accumulate chain on one register needs no wait states.  PART selects a contiguous piece of that
// sequence (0: all eight, 1/2: first / second four, 3..6: the pairs) so that operand loads can be
// issued BETWEEN the MFMAs of a group.
template <int RI, int ROT, int PART = 0>
__device__ __forceinline__ void group(const sl_d2& av, const BFrag& b) {
    constexpr int N0 = 2 * ((RI * CB + 0) * 4 + ROT), N1 = 2 * ((RI * CB + 1) * 4 + ROT);
    constexpr int N2 = 2 * ((RI * CB + 2) * 4 + ROT), N3 = 2 * ((RI * CB + 3) * 4 + ROT);
#define SL_GP4_X01 "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %0, %2, a[%c10:%c11]\n\t" \
                   "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %0, %4, a[%c12:%c13]"
#define SL_GP4_X23 "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %0, %6, a[%c14:%c15]\n\t" \
                   "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %0, %8, a[%c16:%c17]"
#define SL_GP4_Y01 "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %1, %3, a[%c10:%c11]\n\t" \
                   "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %1, %5, a[%c12:%c13]"
#define SL_GP4_Y23 "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %1, %7, a[%c14:%c15]\n\t" \
                   "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %1, %9, a[%c16:%c17]"
// The clobber list names the accumulators the piece writes (a[32 RI + 16 h ...], sl_gp4_clobbers.h).
// Listing the whole file on every statement made the compiler separate adjacent statements with
// an s_nop (write-after-write on the clobbered registers) - one bubble per group.  That the
// compiler keeps no value of its own in ANY accumulator register is established by the kernel's
// opening statement, -amdgpu-spill-vgpr-to-agpr=0 and the build-time audit of the generated code.
#define SL_GP4_ASM_CL(TEXT, ...)                                                                   \
    asm volatile(TEXT                                                                              \
                 :                                                                                 \
                 : "v"(av.x), "v"(av.y), "v"(b.v[0].x), "v"(b.v[0].y), "v"(b.v[1].x), "v"(b.v[1].y), \
                   "v"(b.v[2].x), "v"(b.v[2].y), "v"(b.v[3].x), "v"(b.v[3].y), "i"(N0), "i"(N0 + 1), \
                   "i"(N1), "i"(N1 + 1), "i"(N2), "i"(N2 + 1), "i"(N3), "i"(N3 + 1)                 \
                 : __VA_ARGS__)
#define SL_GP4_ASM_R(TEXT, R_, H_) SL_GP4_ASM_CL(TEXT, SL_GP4_CL_##R_##H_)
#define SL_GP4_ASM_H(TEXT, H_)                                     \
    do {                                                           \
        if constexpr (RI == 0) SL_GP4_ASM_R(TEXT, 0, H_);          \
        else if constexpr (RI == 1) SL_GP4_ASM_R(TEXT, 1, H_);     \
        else if constexpr (RI == 2) SL_GP4_ASM_R(TEXT, 2, H_);     \
        else if constexpr (RI == 3) SL_GP4_ASM_R(TEXT, 3, H_);     \
        else if constexpr (RI == 4) SL_GP4_ASM_R(TEXT, 4, H_);     \
        else if constexpr (RI == 5) SL_GP4_ASM_R(TEXT, 5, H_);     \
        else if constexpr (RI == 6) SL_GP4_ASM_R(TEXT, 6, H_);     \
        else SL_GP4_ASM_R(TEXT, 7, H_);                            \
    } while (0)
    // H_: empty = both cell-block pairs, _0 / _1 = cell blocks 0-1 / 2-3
    if constexpr (PART == 0) SL_GP4_ASM_H(SL_GP4_X01 "\n\t" SL_GP4_X23 "\n\t" SL_GP4_Y01 "\n\t" SL_GP4_Y23, );
    else if constexpr (PART == 1) SL_GP4_ASM_H(SL_GP4_X01 "\n\t" SL_GP4_X23, );
    else if constexpr (PART == 2) SL_GP4_ASM_H(SL_GP4_Y01 "\n\t" SL_GP4_Y23, );
    else if constexpr (PART == 3) SL_GP4_ASM_H(SL_GP4_X01, _0);
    else if constexpr (PART == 4) SL_GP4_ASM_H(SL_GP4_X23, _1);
    else if constexpr (PART == 5) SL_GP4_ASM_H(SL_GP4_Y01, _0);
    else SL_GP4_ASM_H(SL_GP4_Y23, _1);
#undef SL_GP4_ASM_H
#undef SL_GP4_ASM_R
#undef SL_GP4_ASM_CL
#undef SL_GP4_X01
#undef SL_GP4_X23
#undef SL_GP4_Y01
#undef SL_GP4_Y23
}
// One rotation of a slab pair against the row blocks r >= R0, operand loads INSIDE the MFMA
// stream: with one wavefront per SIMD every instruction issued between two groups is a bubble of
// the matrix pipe (four ds_read_b128 in a row: 20-30 cycles per 130-cycle group), issued between
// the MFMAs of a group it disappears in the 16-cycle shadow of the previous MFMA.  The first
// group of the rotation carries the four k_x fragment reads of the NEXT rotation (two after each of
// its first two MFMA pairs: at least four MFMAs old when the next rotation starts, also when the
// rotation is a single group); group r carries the
// buffer load of A fragment r of the slab pair two ahead in the rotation (r - R0) & 3, so every
// fragment is requested once per slab pair.  sched_barrier pins the placement (the scheduler
// would otherwise gather the loads in front of the group).
template <int R0, int ROT, bool LOADA, int S2, int RI = R0, int REND = R>
__device__ __forceinline__ void rotation(const AFrag& a, const BFrag& b, BFrag& bn, const double* kxn,
                                         AFrag& an, __amdgpu_buffer_rsrc_t rsrc,
                                         const int (&rowoff)[R], int s2n, int lane) {
    if constexpr (RI < REND) {
        constexpr bool LA = LOADA && ((RI - R0) & 3) == ROT;
        if constexpr (RI == R0) {
            group<RI, ROT, 3>(a.v[RI], b);
            bn.v[0] = *reinterpret_cast<const sl_d2*>(kxn);
            bn.v[1] = *reinterpret_cast<const sl_d2*>(kxn + 128);
            __builtin_amdgcn_sched_barrier(0);
            group<RI, ROT, 4>(a.v[RI], b);
            bn.v[2] = *reinterpret_cast<const sl_d2*>(kxn + 256);
            bn.v[3] = *reinterpret_cast<const sl_d2*>(kxn + 384);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LA) {
                group<RI, ROT, 5>(a.v[RI], b);
                an.v[RI] = __builtin_bit_cast(
                    sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rowoff[RI] + s2n * 1024, 0));
                __builtin_amdgcn_sched_barrier(0);
                group<RI, ROT, 6>(a.v[RI], b);
            } else {
                group<RI, ROT, 2>(a.v[RI], b);
            }
        } else if constexpr (LA) {
            group<RI, ROT, 1>(a.v[RI], b);
            an.v[RI] = __builtin_bit_cast(
                sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rowoff[RI] + s2n * 1024, 0));
            __builtin_amdgcn_sched_barrier(0);
            group<RI, ROT, 2>(a.v[RI], b);
        } else {
            group<RI, ROT>(a.v[RI], b);
        }
        rotation<R0, ROT, LOADA, S2, RI + 1, REND>(a, b, bn, kxn, an, rsrc, rowoff, s2n, lane);
    }
}
__device__ __forceinline__ void load_b(BFrag& b, const double* kxs, int off) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b.v[cb] = *reinterpret_cast<const sl_d2*>(kxs + cb * 128 + off);
}
// A fragments of slab pair s2abs for the row blocks r >= R0: one coalesced 1 KiB buffer load per
// fragment, the fragment's byte offset in the scalar operand
template <int R0>
__device__ __forceinline__ void load_a(AFrag& a, __amdgpu_buffer_rsrc_t rsrc, const int (&rowoff)[R],
                                       int s2abs, int lane) {
#pragma unroll
    for (int r = R0; r < R; ++r)
        a.v[r] = __builtin_bit_cast(
            sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rowoff[r] + s2abs * 1024, 0));
}
// the slab pairs S2 .. 7 of a chunk, unrolled: four
// rotations each; `be` holds the fragments of rotation 0 on entry and those of the next slab
// pair's rotation 0 on exit; A fragments two slab pairs ahead in three register sets
template <int R0, int S2>
__device__ __forceinline__ void slab_pairs(AFrag (&a)[3], BFrag& be, BFrag& bo, const double* kxb,
                                           const int (&boff)[4], __amdgpu_buffer_rsrc_t rsrc,
                                           const int (&rowoff)[R], int ch, int lane) {
    if constexpr (S2 < 8) {
        constexpr bool LOADA = S2 < 6;
        const double* kxs = kxb + S2 * KXS2;
        const double* kxs_next = kxb + (S2 < 7 ? S2 + 1 : 7) * KXS2;
        const AFrag& ac = a[S2 % 3];
        AFrag& an = a[(S2 + 2) % 3];
        const int s2n = 8 * ch + S2 + 2;
        rotation<R0, 0, LOADA, S2>(ac, be, bo, kxs + boff[1], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 1, LOADA, S2>(ac, bo, be, kxs + boff[2], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 2, LOADA, S2>(ac, be, bo, kxs + boff[3], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 3, LOADA, S2>(ac, bo, be, kxs_next + boff[0], an, rsrc, rowoff, s2n, lane);
        slab_pairs<R0, S2 + 1>(a, be, bo, kxb, boff, rsrc, rowoff, ch, lane);
    }
}
// The DIAGONAL block of a wavefront (row block R0 in chunk R0 of the panel's diagonal band): its 16
// rows end inside the chunk, so its fragments are zero from slab pair `nz` on (2, 4, 6 or 8: the
// wavefront that owns block s of the band's four has rows 16 s .. 16 s + 15 of the chunk's 64
// columns), and a product with those zeros leaves the accumulators as they are.  The diagonal block
// therefore runs on its own, in front of the blocks below it, and stops after `nz` slab pairs (a
// wave-uniform test in front of every second one).  At n = 1024, 6 of the 136 (row block, chunk)
// products of a tile are such zeros: 4.4 % of the MFMAs; measured 3.0 % of the sweep at 64^4
// (profiles/r05_gp4_diag_ab.txt: the split itself, never leaving early, costs 0.5 %).
template <int R0, int S2>
__device__ __forceinline__ void diag_pairs(AFrag (&a)[3], BFrag& be, BFrag& bo, const double* kxb,
                                           const int (&boff)[4], __amdgpu_buffer_rsrc_t rsrc,
                                           const int (&rowoff)[R], int ch, int lane, int nz) {
    if constexpr (S2 < 8) {
        if constexpr (S2 >= 2 && S2 % 2 == 0) {
            if (S2 == nz) return;
        }
        constexpr bool LOADA = S2 < 6;
        const double* kxs = kxb + S2 * KXS2;
        const double* kxs_next = kxb + (S2 < 7 ? S2 + 1 : 7) * KXS2;
        const AFrag& ac = a[S2 % 3];
        AFrag& an = a[(S2 + 2) % 3];
        const int s2n = 8 * ch + S2 + 2;
        rotation<R0, 0, LOADA, S2, R0, R0 + 1>(ac, be, bo, kxs + boff[1], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 1, LOADA, S2, R0, R0 + 1>(ac, bo, be, kxs + boff[2], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 2, LOADA, S2, R0, R0 + 1>(ac, be, bo, kxs + boff[3], an, rsrc, rowoff, s2n, lane);
        rotation<R0, 3, LOADA, S2, R0, R0 + 1>(ac, bo, be, kxs_next + boff[0], an, rsrc, rowoff, s2n, lane);
        diag_pairs<R0, S2 + 1>(a, be, bo, kxb, boff, rsrc, rowoff, ch, lane, nz);
    }
}
// one chunk of 64 training points against the row blocks r >= R0 (DIAG: R0 is the diagonal block)
template <int R0, bool DIAG>
__device__ __forceinline__ void chunk(__amdgpu_buffer_rsrc_t rsrc, const double* kxb,
                                      const int (&rowoff)[R], int ch, int lane, const int (&boff)[4],
                                      int nz) {
    AFrag a[3];
    BFrag be, bo;
    if constexpr (DIAG) {
        // the diagonal block first; the first fragments of the blocks below it are requested with
        // its own (one trip to L2 in front of the chunk, as without the split)
        AFrag ad[3];
        ad[0].v[R0] = __builtin_bit_cast(
            sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rowoff[R0] + (8 * ch) * 1024, 0));
        ad[1].v[R0] = __builtin_bit_cast(
            sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rowoff[R0] + (8 * ch + 1) * 1024, 0));
        if constexpr (R0 + 1 < R) {
            load_a<R0 + 1>(a[0], rsrc, rowoff, 8 * ch, lane);
            load_a<R0 + 1>(a[1], rsrc, rowoff, 8 * ch + 1, lane);
        }
        load_b(be, kxb, boff[0]);
        diag_pairs<R0, 0>(ad, be, bo, kxb, boff, rsrc, rowoff, ch, lane, nz);
        if constexpr (R0 + 1 < R) {
            // The blocks below READ their first k_x fragments again.  (Handing them over from the
            // diagonal stream - its last rotation fetching slab pair 0 - made the compiler join the
            // four ways out of that stream with register copies right in front of the first MFMA:
            // a VALU write of an MFMA source one issue slot ahead of the MFMA, which the hardware
            // does not interlock for the inline-asm MFMAs - wrong |a|^2, found by the parity tests;
            // tools/audit_gp4.py now refuses such a listing.)
            load_b(be, kxb, boff[0]);
            asm volatile("s_nop 1");
            slab_pairs<R0 + 1, 0>(a, be, bo, kxb, boff, rsrc, rowoff, ch, lane);
        }
    } else {
        load_a<R0>(a[0], rsrc, rowoff, 8 * ch, lane);
        load_a<R0>(a[1], rsrc, rowoff, 8 * ch + 1, lane);
        load_b(be, kxb, boff[0]);
        slab_pairs<R0, 0>(a, be, bo, kxb, boff, rsrc, rowoff, ch, lane);
    }
}
// q = chunk index relative to the panel's diagonal band.  Row block r of a wavefront has its
// diagonal in chunk r of the band: blocks r >= q are active (the diagonal block's fragments are
// zero above the diagonal: from slab pair nzd[q] on), blocks r < q lie above it; q < 0: every block
// is active and full.
__device__ __forceinline__ void chunk_any(__amdgpu_buffer_rsrc_t rsrc, const double* kxb,
                                          const int (&rowoff)[R], int q, int ch, int lane,
                                          const int (&boff)[4], const int (&nzd)[R]) {
    switch (q) {
        case 0: chunk<0, true>(rsrc, kxb, rowoff, ch, lane, boff, nzd[0]); break;
        case 1: chunk<1, true>(rsrc, kxb, rowoff, ch, lane, boff, nzd[1]); break;
        case 2: chunk<2, true>(rsrc, kxb, rowoff, ch, lane, boff, nzd[2]); break;
        case 3: chunk<3, true>(rsrc, kxb, rowoff, ch, lane, boff, nzd[3]); break;
        default: chunk<0, false>(rsrc, kxb, rowoff, ch, lane, boff, 8); break;
    }
}

// exp of two arguments (sl_exp_nonpos twice), the two dependent FMA chains written alternately: a
// single wavefront per SIMD has nobody else to fill the latency of a 13-deep chain
__device__ __forceinline__ void exp_pair(double x1, double x2, double& e1, double& e2) {
    x1 = x1 < -800.0 ? -800.0 : x1;
    x2 = x2 < -800.0 ? -800.0 : x2;
    const double k1 = rint(x1 * 1.4426950408889634), k2 = rint(x2 * 1.4426950408889634);
    double r1 = fma(k1, -6.93147180369123816490e-01, x1), r2 = fma(k2, -6.93147180369123816490e-01, x2);
    r1 = fma(k1, -1.90821492927058770002e-10, r1);
    r2 = fma(k2, -1.90821492927058770002e-10, r2);
    double q1 = 1.6059043836821613e-10, q2 = 1.6059043836821613e-10;
#define SL_EXP_STEP(C) q1 = fma(q1, r1, C); q2 = fma(q2, r2, C)
    SL_EXP_STEP(2.08767569878681e-09);
    SL_EXP_STEP(2.505210838544172e-08);
    SL_EXP_STEP(2.755731922398589e-07);
    SL_EXP_STEP(2.7557319223985893e-06);
    SL_EXP_STEP(2.48015873015873e-05);
    SL_EXP_STEP(1.984126984126984e-04);
    SL_EXP_STEP(1.3888888888888889e-03);
    SL_EXP_STEP(8.333333333333333e-03);
    SL_EXP_STEP(4.1666666666666664e-02);
    SL_EXP_STEP(1.6666666666666666e-01);
    SL_EXP_STEP(0.5);
    SL_EXP_STEP(1.0);
    SL_EXP_STEP(1.0);
#undef SL_EXP_STEP
    e1 = ldexp(q1, (int)k1);
    e2 = ldexp(q2, (int)k2);
}

// a wave-uniform double held in scalar registers
__device__ __forceinline__ double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

}  // namespace gp4

// (The scaled training inputs and alpha' are read from L2 through buffer resources: two workgroups
// per CU leave no LDS for them beside the k_x buffers.)
template <int DT, int MT>
__global__ __launch_bounds__(256, 2) void k_gp_sweep4(
    const SlDevModel M, const SlGpDev gp, SlAux aux, int64_t lo, int64_t hi, int64_t ntiles,
    const uint64_t* __restrict__ init_bits, const double* __restrict__ values,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    const double* __restrict__ points, int skip_arg, double* __restrict__ seeds, int seed_chunks,
    unsigned long long* __restrict__ ticket) {
    // skip (SL_GP4_SKIP; development builds, -DSL_DIAG, only): 1 no k_x generation, 2 no mean pass,
    // 4 no per-cell check, 8 no MFMA chunks - timing attribution, the results are then meaningless.
    // The shipped kernel has no such switch: the tests fold to constants.
#ifdef SL_DIAG
    const int skip = skip_arg;
#else
    constexpr int skip = 0;
    (void)skip_arg;
#endif
    using namespace gp4;
    asm volatile("" ::: SL_ALL_AGPRS);           // the accumulator file belongs to the MFMA groups
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* kx_l = smem;                           // [2][KXBUF]
    constexpr int PSS = W;                         // planes of |a|^2 partials: one per wavefront
    double* part_ss = kx_l + 2 * KXBUF;            // [W][C]
    double* cell_mean = part_ss + PSS * C;         // [C][SL_D]
    double* cell_err = cell_mean + C * SL_D;       // [C][SL_D]
    double* cin = cell_err + C * SL_D;             // [C][SL_P] scaled GP inputs of the tile's cells
    double* runc = cin + C * SL_P;                 // [W][RUNS][RUNC] step, a^2, Q of a wavefront's runs
    uint64_t* sv = reinterpret_cast<uint64_t*>(runc + W * RUNS * RUNC);   // [W]
    int64_t* si = reinterpret_cast<int64_t*>(sv + W);                  // [W]
    int64_t* next_tile = si + W;                                       // the tile the workgroup drew

    const SlDims nd = sl_dims<DT, MT>(M);
    const int d = nd.d, p = nd.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // = cell block this wave generates
    const int lcol = lane & 15, lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int boff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
        boff[rot] = 2 * (16 * lk + ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15));
    const int own = 2 * (16 * lk + ((lcol + 4 * (lk >> 1)) & 15));  // this lane's own (k, cell) item
    // generation writes (lane = point jj of the chunk): slab pair jj >> 3, slab (jj >> 2) & 1, k = jj & 3
    const int wbase = (lane >> 3) * KXS2 + wave * 128 + 32 * (lane & 3) + ((lane >> 2) & 1);
    const int wswz = 4 * ((lane & 3) >> 1);
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    // Seeds of the Gaussian sequences, [run][chunk][wavefront][lane][e_0, rho_0] per workgroup: a
    // panel of 256 rows walks over every chunk of the panels before it again (40 chunk generations
    // per tile for 16 different chunks at n = 1024).  The first generation of a chunk keeps
    // (e_0, rho_0) - the two exponentials and the distance sums, two thirds of its instructions -
    // in this scratch and the later ones only run the recurrence from them: the same lane reads
    // back its own 16 bytes, the products are the same products, the k_x values bit for bit the
    // same.  (Measured: 0.5 % of the sweep, profiles/r04_summary.md section 2.)  Tiles with kinks
    // keep one pair per run.  Round 5: the workgroup's slice is a buffer resource and a seed pair
    // is addressed by a scalar offset (chunk, run, wavefront) + 16 lane: no 64-bit pointer per lane
    // to carry through (or spill around) the MFMA streams; the same for the training inputs and
    // alpha' below.
    const bool use_seeds = seeds != nullptr;
    const int seed_run_bytes = seed_chunks * W * 1024;          // one run's seeds of this workgroup
    __amdgpu_buffer_rsrc_t rs_seed = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(seeds + (size_t)blockIdx.x * RUNS * seed_chunks * W * 128), 0, 0x7fffffff, 0x27000);
    const int seed_wave = wave * 1024;

    // Tiles are drawn from a counter: a tile whose cells cross a saturation kink of the policy
    // generates its k_x chunks at twice the cost (a fifth of the headline workload's tiles), and
    // with a fixed tile list per workgroup the unlucky workgroups finish last (measured: 2 % of
    // the sweep, profiles/r04_gp4_tickets_ab.txt).
    // (Few tiles per workgroup - small grids, C2's 1024 tiles on 512 workgroups: the counter hands
    // one workgroup three tiles and another one, and the sweep lasts as long as the three; a fixed
    // round-robin list is the better balance there: ticket == nullptr.)
    for (int64_t round = 0;; ++round) {
        if (tid == 0) *next_tile = ticket ? (int64_t)atomicAdd(ticket, 1ull) : (int64_t)blockIdx.x + round * gridDim.x;
        __syncthreads();
        const int64_t tile = *next_tile;       // rewritten after the barriers of the tile
        if (tile >= ntiles) break;
        const int64_t tile_base = lo + tile * C;
        if (tile_base >= hi) {                     // padding tile: only clears mask bits
            if (tid == 0) neg_bits[(tile_base - lo) >> 6] = 0ull;
            __syncthreads();                       // everybody has read next_tile
            continue;
        }

        for (int h = 0; h < gp.nheads; ++h) {
            const SlGpHeadDev& hd = gp.head[h];
            const int n_pad = hd.n_pad, dout = hd.dout;
            const double variance = hd.variance;
            __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)hd.mpack, 0, 0x7fffffff, 0x27000);
            __amdgpu_buffer_rsrc_t rs_xs =
                __builtin_amdgcn_make_buffer_rsrc((void*)hd.xs, 0, 0x7fffffff, 0x27000);
            __amdgpu_buffer_rsrc_t rs_alpha =
                __builtin_amdgcn_make_buffer_rsrc((void*)hd.alpha, 0, 0x7fffffff, 0x27000);
            // scaled GP input [x, policy(x)] / lengthscales of the 64 cells (lane = cell of block `wave`)
            if (lk == 0) {
                double xg[SL_P], u[SL_M];
                int64_t gidx = tile_base + 16 * wave + lcol;
                gidx = gidx < hi ? gidx : hi - 1;
                // (opaque: the decode of the index stays here - hoisted out of the loop over the
                // heads its results were spilled to scratch at every tile)
                asm volatile("" : "+v"(gidx));
                sl_cell_state(M, d, gidx, points, xg);
                sl_policy_any<false>(M, nd, aux.tri, gidx, xg, u);
                sl_append_action(nd, u, xg);
#pragma unroll
                for (int q = 0; q < SL_P; ++q)
                    cin[(16 * wave + lcol) * SL_P + q] = (q < p) ? xg[q] * hd.inv_ls[q] : 0.0;
            }
            __syncthreads();
            // Where the input is affine in the cell index, z_j(c) = |X_j - x(c)|^2 is quadratic in
            // c and k_x a Gaussian sequence.  Split the wavefront's 16 cells into maximal affine
            // runs (second differences vanish inside a run; a saturation kink or the end of a grid
            // row starts a new one): bit c of `runs` = cell c starts a run.  More than four runs
            // (curved policies, explicit points): one exponential per (point, cell) instead.
            unsigned runs = 1u;
            bool wide = false;
            {
                bool bend = false;
                if (lcol >= 1) {                    // |step|^2 of the scaled input between neighbours
                    double st2 = 0.0;
#pragma unroll
                    for (int q = 0; q < SL_P; ++q) {
                        if (q < p) {
                            const double st = cin[(16 * wave + lcol) * SL_P + q] -
                                              cin[(16 * wave + lcol - 1) * SL_P + q];
                            st2 = fma(st, st, st2);
                        }
                    }
                    wide = !(st2 <= 1.0);           // also NaN
                }
                if (lcol >= 2) {
#pragma unroll
                    for (int q = 0; q < SL_P; ++q) {
                        if (q < p) {
                            const double c0 = cin[(16 * wave + lcol) * SL_P + q];
                            const double c1 = cin[(16 * wave + lcol - 1) * SL_P + q];
                            const double c2 = cin[(16 * wave + lcol - 2) * SL_P + q];
                            bend = bend || fabs((c0 - c1) - (c1 - c2)) > 2e-14 * fmax(1.0, fabs(c0));
                        }
                    }
                }
                unsigned bends = (unsigned)(__ballot(bend) & 0xffffull);
                // a bend at c starts a run at c; c + 1 is then its second point, not a new start
                while (bends) {
                    const int c = __builtin_ctz(bends);
                    runs |= 1u << c;
                    bends &= ~(3u << c);
                }
            }
            runs = (unsigned)__builtin_amdgcn_readfirstlane((int)runs);
            // The recurrence forms e_0 = s^2 exp(-z/2) and rho_0 = exp(b - a^2/2) separately: with
            // steps longer than a lengthscale (a^2 > 1) e_0 can underflow for a training point
            // that a later cell of the run comes close to, so such tiles take one exponential
            // per (point, cell) like curved inputs do.  With a^2 <= 1 a point whose e_0 underflows
            // stays > 20 lengthscales away from all 16 cells (true value < 1e-100), and the
            // exponent of rho is capped (b > 700 implies e_0 = 0 exactly: 0 * finite, not 0 * inf).
            const bool direct = __builtin_popcount(runs) > 4 || (__ballot(wide) & 0xffffull) != 0ull;
            // constants of the first run (the only one for most tiles) in scalar registers
            double x0[SL_P], dlt[SL_P], a2 = 0.0;
#pragma unroll
            for (int q = 0; q < SL_P; ++q) {
                if (q < p) {
                    x0[q] = uniform(cin[(16 * wave) * SL_P + q]);
                    dlt[q] = uniform(cin[(16 * wave + 1) * SL_P + q] - x0[q]);
                    a2 = fma(dlt[q], dlt[q], a2);
                }
            }
            a2 = uniform(a2);
            const double qstep = uniform(sl_exp_nonpos(-a2));
            // a tile with a kink: the step, |step|^2 and Q of every run once per tile (lane r
            // for run r), not once per training point and chunk
            if (runs != 1u && !direct) {
                if (lane < RUNS) {
                    unsigned m = runs;
                    for (int r = 0; r < lane; ++r) m &= m - 1;
                    if (m) {
                        const int c0 = __builtin_ctz(m);
                        m &= m - 1;
                        const int c1 = m ? __builtin_ctz(m) : 16;
                        const double* at = cin + (16 * wave + c0) * SL_P;
                        const bool single = c1 - c0 < 2;
                        double a2r = 0.0;
                        double* rc = runc + (wave * RUNS + lane) * RUNC;
#pragma unroll
                        for (int q = 0; q < SL_P; ++q) {
                            const double step = (q < p && !single) ? at[SL_P + q] - at[q] : 0.0;
                            rc[q] = step;
                            a2r = fma(step, step, a2r);
                        }
                        rc[SL_P] = a2r;
                        rc[SL_P + 1] = sl_exp_nonpos(-a2r);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            double macc[4] = {0.0, 0.0, 0.0, 0.0};   // posterior-mean accumulators (see mean_pass)

            // k_x chunk `ch` -> LDS buffer `buf` (lane = training point 64 ch + lane)
            // first_new: chunks below it were generated for an earlier panel of this tile;
            // keep: a later panel will want this chunk again
            auto generate = [&](int ch, int buf, int first_new, bool keep) {
                double* kxw = kx_l + buf * KXBUF + wbase;
                double xv[SL_P];
                auto load_xv = [&]() {
#pragma unroll
                    for (int q = 0; q < SL_P; ++q)
                        if (q < p)
                            xv[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                rs_xs, lane * 8, (q * n_pad + 64 * ch) * 8, 0));
                };
                if (runs == 1u) {                  // one affine run: e_{c+1} = e_c rho_c, rho_{c+1} = rho_c Q
                    double e, rho;
                    if (use_seeds && ch < first_new) {
                        const sl_d2 sd = __builtin_bit_cast(sl_d2, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_seed, lane * 16, ch * (W * 1024) + seed_wave, 0));
                        e = sd.x;
                        rho = sd.y;
                    } else {
                        load_xv();
                        double z = 0.0, bj = 0.0;
#pragma unroll
                        for (int q = 0; q < SL_P; ++q) {
                            if (q < p) {
                                const double dq = xv[q] - x0[q];
                                z = fma(dq, dq, z);
                                bj = fma(dq, dlt[q], bj);
                            }
                        }
                        exp_pair(-0.5 * z, fmin(bj - 0.5 * a2, 700.0), e, rho);
                        e = variance * e;
                        if (use_seeds && keep) {
                            sl_d2 sd;
                            sd.x = e;
                            sd.y = rho;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sl_u4, sd), rs_seed, lane * 16,
                                                                   ch * (W * 1024) + seed_wave, 0);
                        }
                    }
                    // slot (c + wswz) & 15 with wswz 0 or 4: two bases, immediate offsets
                    double* w_lo = kxw + 2 * wswz;               // cells 0..11
                    double* w_hi = w_lo - 8 * wswz;              // cells 12..15 wrap for wswz = 4
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        (c < 12 ? w_lo : w_hi)[2 * c] = e;
                        e *= rho;
                        rho *= qstep;
                    }
                } else if (!direct) {              // a few runs: the recurrence restarts at each
                    const bool reuse = use_seeds && ch < first_new;
                    if (!reuse) load_xv();
                    unsigned m = runs;
                    for (int r = 0; m; ++r) {
                        const int c0 = __builtin_ctz(m);
                        m &= m - 1;
                        const int c1 = m ? __builtin_ctz(m) : 16;
                        const double* at = cin + (16 * wave + c0) * SL_P;
                        const double* rc = runc + (wave * RUNS + r) * RUNC;
                        // run r of a chunk: its seeds sit one run's seeds behind those of run r - 1
                        const int sp = r * seed_run_bytes + ch * (W * 1024) + seed_wave;
                        double e, rho;
                        if (reuse) {
                            const sl_d2 sd = __builtin_bit_cast(sl_d2, __builtin_amdgcn_raw_buffer_load_b128(
                                rs_seed, lane * 16, sp, 0));
                            e = sd.x;
                            rho = sd.y;
                        } else {
                            double z = 0.0, bj = 0.0;
#pragma unroll
                            for (int q = 0; q < SL_P; ++q) {
                                if (q < p) {
                                    const double dq = xv[q] - at[q];
                                    z = fma(dq, dq, z);
                                    bj = fma(dq, rc[q], bj);
                                }
                            }
                            exp_pair(-0.5 * z, fmin(bj - 0.5 * rc[SL_P], 700.0), e, rho);
                            e = variance * e;
                            if (use_seeds && keep) {
                                sl_d2 sd;
                                sd.x = e;
                                sd.y = rho;
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sl_u4, sd), rs_seed,
                                                                       lane * 16, sp, 0);
                            }
                        }
                        const double qr = rc[SL_P + 1];
                        for (int c = c0; c < c1; ++c) {
                            kxw[2 * ((c + wswz) & 15)] = e;
                            e *= rho;
                            rho *= qr;
                        }
                    }
                } else {
                    load_xv();
                    for (int c = 0; c < 16; ++c) {
                        double z = 0.0;
#pragma unroll
                        for (int q = 0; q < SL_P; ++q) {
                            if (q < p) {
                                const double dq = xv[q] - cin[(16 * wave + c) * SL_P + q];
                                z = fma(dq, dq, z);
                            }
                        }
                        kxw[2 * ((c + wswz) & 15)] = variance * sl_exp_nonpos(-0.5 * z);
                    }
                }
            };
            // Posterior mean k_x . alpha' on the matrix cores as well: for the wavefront's own 16
            // cells, mean[dd][cell] = sum_j alpha'[j][dd] k_x[j][cell] is a 4 x 16 x 64 product per
            // chunk - A = alpha'^T (row dd = lane & 3, the same for the four blocks), B = the
            // rotation-0 k_x fragment of cell block `wave`.  Lane (k, blk, low) ends up with the
            // mean of output dd = k at cell 4 blk + low.  (As FP64-VALU work in the (k, cell)
            // lane mapping this cost 8 % of the sweep.)
            // The pass comes in two parts around the generation of the same chunk: alpha' is
            // requested together with the inputs of the generation (one trip to L2 for
            // both), mean_run multiplies once the wavefront has written its k_x values - its
            // OWN values (lane = training point, the 16 cells of block `wave`): LDS serves a
            // wavefront's instructions in order, no workgroup barrier is needed in between.
            struct MeanIn { double a0[8], a1[8]; };
            auto mean_run = [&](int buf, const MeanIn& mi) {
                const double* kxr = kx_l + buf * KXBUF + wave * 128 + own;
                sl_d2 kx[8];
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) kx[s2] = *reinterpret_cast<const sl_d2*>(kxr + s2 * KXS2);
                // Accumulators in vector registers (the builtin would route them through a0:a1).
                // A dependent FP64 MFMA must not issue right behind its producer (no interlock:
                // measured, the second product was lost): four accumulators in rotation keep
                // three MFMAs between a write and its reuse.
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile("s_nop 3\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %1, %6, %7, %1\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %3, %10, %11, %3\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %0, %12, %13, %0\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %1, %14, %15, %1\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %2, %16, %17, %2\n\t"
                                 "v_mfma_f64_4x4x4_4b_f64 %3, %18, %19, %3"
                                 : "+v"(macc[0]), "+v"(macc[1]), "+v"(macc[2]), "+v"(macc[3])
                                 : "v"(mi.a0[4 * h]), "v"(kx[4 * h].x), "v"(mi.a1[4 * h]), "v"(kx[4 * h].y),
                                   "v"(mi.a0[4 * h + 1]), "v"(kx[4 * h + 1].x), "v"(mi.a1[4 * h + 1]), "v"(kx[4 * h + 1].y),
                                   "v"(mi.a0[4 * h + 2]), "v"(kx[4 * h + 2].x), "v"(mi.a1[4 * h + 2]), "v"(kx[4 * h + 2].y),
                                   "v"(mi.a0[4 * h + 3]), "v"(kx[4 * h + 3].x), "v"(mi.a1[4 * h + 3]), "v"(kx[4 * h + 3].y));
                // retired before any other reader (a register copy, the final sum)
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(macc[0]), "+v"(macc[1]), "+v"(macc[2]), "+v"(macc[3]));
            };
            // chunk `ch` -> LDS buffer `buf`, and its contribution to the posterior mean
            auto produce = [&](int ch, int buf, int first_new, bool keep) {
                const bool with_mean = ch >= first_new && !(skip & 2);
                MeanIn mi;
                if (with_mean) {
                    // alpha' from L2 through the head's buffer resource: one lane offset, the row
                    // of the slab in the scalar operand.  Rows dd >= dout of A hold whatever a valid
                    // column holds: row dd of the product depends on row dd of A only, and the rows
                    // >= dout of the result are never read.
                    const int voff = (lk * dout + (low < dout ? low : 0)) * 8;
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        mi.a0[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                            rs_alpha, voff, (64 * ch + 8 * s2) * dout * 8, 0));
                        mi.a1[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                            rs_alpha, voff, (64 * ch + 8 * s2 + 4) * dout * 8, 0));
                    }
                }
                if (!(skip & 1)) generate(ch, buf, first_new, keep);
                if (with_mean) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    mean_run(buf, mi);
                }
            };

            // panels that hold training points (the upload pads to the 16x16x4 fallback's 512 rows:
            // a panel of zero rows contributes nothing - 600 points are three panels, not four)
            const int npanels = (hd.n + RP - 1) / RP;
            for (int pan = 0; pan < npanels; ++pan) {
                acc_zero_all();
                int rowoff[R];                     // byte offset of each owned row block's fragments
                int nzd[R];                        // slab pairs of its diagonal chunk that are not zero
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int wsel = (r & 1) ? (W - 1 - wave) : wave;     // balance the triangle
                    rowoff[r] = (pan * RB + r * W + wsel) * hd.nslab2 * 1024;
                    nzd[r] = 2 * wsel + 2;
                }
                constexpr int CPP = RP / 64;                     // chunks per panel (its diagonal band)
                const int nchunks = (pan + 1) * CPP;
                const int first_new_chunk = pan * (RP / 64);     // chunks not generated before
                const bool keep = pan + 1 < npanels;
                produce(0, 0, first_new_chunk, keep);
                __syncthreads();
                for (int ch = 0; ch < nchunks; ++ch) {
                    const int buf = ch & 1;
                    const int q = __builtin_amdgcn_readfirstlane(ch - CPP * pan);
                    const double* kxb = kx_l + buf * KXBUF;
                    if (!(skip & 8)) chunk_any(rsrc, kxb, rowoff, q, ch, lane, boff, nzd);
                    if (ch + 1 < nchunks) produce(ch + 1, buf ^ 1, first_new_chunk, keep);
                    __syncthreads();
                }
                // |a|^2 of this panel's rows.  The rows of a block live in the four lane groups
                // (row = lane >> 4): fold them; every (wave, rotation) plane of part_ss then holds
                // one partial sum per cell, owned by one lane (no other wave touches the plane).
                asm volatile("s_nop 15\n\ts_nop 15" ::: SL_ALL_AGPRS);   // MFMA results -> reads
                double ssr[CB][4];
                acc_gram(ssr);
                // retired before the stores read them
                asm volatile("s_nop 15\n\ts_nop 7"
                             : "+v"(ssr[0][0]), "+v"(ssr[0][1]), "+v"(ssr[0][2]), "+v"(ssr[0][3]),
                               "+v"(ssr[1][0]), "+v"(ssr[1][1]), "+v"(ssr[1][2]), "+v"(ssr[1][3]),
                               "+v"(ssr[2][0]), "+v"(ssr[2][1]), "+v"(ssr[2][2]), "+v"(ssr[2][3]),
                               "+v"(ssr[3][0]), "+v"(ssr[3][1]), "+v"(ssr[3][2]), "+v"(ssr[3][3]));
                const bool owner = lk == low;          // the diagonal of the Gram matrices
                {
                    // one plane per wavefront (LDS is short): the four rotations of a lane belong
                    // to four different cells, and within a rotation the lanes hit distinct cells,
                    // so the rotations are added one after the other (in-order LDS, same wavefront)
                    if (pan == 0) {
                        if (lane < 16) {
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) part_ss[wave * C + 16 * cb + lane] = 0.0;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll
                    for (int rot = 0; rot < 4; ++rot) {
                        if (owner) {
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) {
                                double* slot = part_ss + wave * C + 16 * cb + 4 * ((blk + rot) & 3) + low;
                                *slot = *slot + ssr[cb][rot];
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (lk < dout)
                cell_mean[(16 * wave + 4 * blk + low) * SL_D + hd.col0 + lk] =
                    (macc[0] + macc[1]) + (macc[2] + macc[3]);
            __syncthreads();
            if (tid < C) {
                double sumsq = 0.0;
                for (int k = 0; k < PSS; ++k) sumsq += part_ss[k * C + tid];
                const double var = variance - sumsq;                       // functions.py:451
                const double e = gp.beta * sqrt(var);                      // functions.py:514
                for (int dd = 0; dd < dout; ++dd) cell_err[tid * SL_D + hd.col0 + dd] = e;
            }
            __syncthreads();
        }

        // ---- per-cell decrease check, mask word, failing-cell key (as k_gp_sweep) -------------------
        const int64_t idx = tile_base + tid;
        const bool valid = (tid < C) && (idx < hi);
        bool negative = false;
        double v_x = 0.0;
        if (valid && !(skip & 4)) {
            double x[SL_P], u[SL_M], prior[SL_D], mean[SL_D], err[SL_D];
            sl_cell_state(M, d, idx, points, x);
            sl_policy_any<false>(M, nd, aux.tri, idx, x, u);
            sl_append_action(nd, u, x);
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);   // m(x*), functions.py:439
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d) {
                    mean[k] = cell_mean[tid * SL_D + k] + prior[k];
                    err[k] = cell_err[tid * SL_D + k];
                }
            }
            SlCellCheck c = sl_cell_check<SL_FAST>(M, d, aux, x, mean, err);
            negative = c.negative;
            v_x = values ? values[idx - lo] : c.v_x;       // ordering key: lyapunov.py:512
            if (dbg) {
                double* o = dbg + (idx - lo) * (2 + 2 * d);
                o[0] = c.decrease; o[1] = c.threshold;
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d) { o[2 + k] = mean[k]; o[2 + d + k] = err[k]; }
            }
        }
        if (wave == 0) {
            const uint64_t word = __ballot(negative);
            uint64_t init = 0ull;
            if (init_bits) init = init_bits[(tile_base - lo) >> 6];
            if (lane == 0) neg_bits[(tile_base - lo) >> 6] = word;
            const bool okc = negative || ((init >> lane) & 1ull);
            if (valid && !okc) sl_key_min(best_v, best_i, sl_vbits(v_x), idx);
        }
        // (cell_mean / cell_err / cin are rewritten only after the next tile's barriers)
    }
    __syncthreads();
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (tid == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// =============================================================================================
// host side
// =============================================================================================
static size_t gp4_fixed_lds() {
    return sizeof(double) * (2 * gp4::KXBUF + gp4::W * gp4::C +
                             2 * gp4::C * SL_D + gp4::C * SL_P + gp4::W * gp4::RUNS * gp4::RUNC) +
           (2 * gp4::W + 2) * sizeof(uint64_t);
}

template <int DT, int MT>
static int launch4(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                   const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                   int* nblocks, double* d_dbg, const double* d_points) {
    const int64_t ntiles = (hi - lo + 63) / 64;
    const int p = model.in_dim;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) {
        if (ctx->gp_heads[h].p != p)
            return sl_fail(ctx, SL_ERR_INVALID, "GP head %d has input dim %d, model has %d", h,
                           ctx->gp_heads[h].p, p);
        if (ctx->gp_heads[h].n_pad % gp4::RP)
            return sl_fail(ctx, SL_ERR_INVALID, "GP head %d: n_pad %d is not a multiple of %d", h,
                           ctx->gp_heads[h].n_pad, gp4::RP);
    }
    const size_t lds = gp4_fixed_lds();                       // two workgroups per CU: < 80 KB each
    static_assert(sizeof(double) * (2 * gp4::KXBUF + gp4::W * gp4::C + 2 * gp4::C * SL_D + gp4::C * SL_P +
                                    gp4::W * gp4::RUNS * gp4::RUNC) + (2 * gp4::W + 2) * sizeof(uint64_t)
                      <= 80 * 1024, "two workgroups of k_gp_sweep4 per CU");
    auto kern = k_gp_sweep4<DT, MT>;
    SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t resident = (int64_t)ctx->num_cu * 2;
    int64_t blocks = ntiles < resident ? ntiles : resident;
    if (blocks > SL_MAX_GRID) blocks = SL_MAX_GRID;
    *nblocks = (int)blocks;
    SlAux aux{ctx->d_tri, ctx->d_net};
    const int skip = sl_diag_flags("SL_GP4_SKIP");            // (development builds only: 0 otherwise)
    // scratch of the sequence seeds: [workgroup][chunk][wavefront][lane][2] (see the kernel)
    int seed_chunks = 0;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) {
        const int c = ctx->gp_heads[h].n_pad / 64;
        seed_chunks = c > seed_chunks ? c : seed_chunks;
    }
    const size_t head_bytes = 16;                               // the tile counter
    const size_t seed_bytes = head_bytes + (size_t)gp4::RUNS * blocks * seed_chunks * gp4::W * 128 * sizeof(double);
    if (seed_bytes > ctx->gp4_seed_bytes) {
        SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_gp4_seeds) (void)hipFree(ctx->d_gp4_seeds);
        ctx->d_gp4_seeds = nullptr;
        ctx->gp4_seed_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_gp4_seeds, seed_bytes));
        ctx->gp4_seed_bytes = seed_bytes;
    }
    // SL_GP4_SEEDS=0: every generation from scratch (same k_x bit for bit: the test of that)
    double* seeds = ctx->env.gp4_seeds == 0 ? nullptr : ctx->d_gp4_seeds + head_bytes / sizeof(double);
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(ctx->d_gp4_seeds);
    // SL_GP4_TICKETS=0 / 1 force the list / the counter
    const bool counter = ctx->env.gp4_tickets >= 0 ? ctx->env.gp4_tickets != 0 : ntiles >= 4 * blocks;
    if (counter) SL_HIP_CHECK(ctx, hipMemsetAsync(ticket, 0, head_bytes, ctx->stream));
    else ticket = nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(gp4::W * 64), lds, ctx->stream, model,
                       ctx->h_gp, aux, lo, hi, ntiles, d_init_bits, d_values, d_neg_bits,
                       ctx->d_partials, d_dbg, d_points, skip, seeds, seed_chunks, ticket);
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_gp_sweep4<d=%d, m=%d> (%d-row panels, %d workgroup(s) per CU)",
                   DT, MT, gp4::RP, 2);
    return SL_OK;
}

// One entry per state dimension.  The build compiles this file once per dimension
// (-DSL_GP4_DIM=1..4, four hipcc jobs side by side: the unrolled chunks make one instantiation a
// minute of compile time); without the macro everything lives in one translation unit.
#define SL_GP4_DIM_ENTRY(D_)                                                                       \
    int sl_gp4_launch_d##D_(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,          \
                            const uint64_t* d_init_bits, const double* d_values,                   \
                            uint64_t* d_neg_bits, int* nblocks, double* d_dbg,                     \
                            const double* d_points) {                                              \
        return launch4<D_, 1>(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,      \
                              d_dbg, d_points);                                                    \
    }
#define SL_GP4_DIM_DECL(D_)                                                                        \
    int sl_gp4_launch_d##D_(sl_ctx*, const SlDevModel&, int64_t, int64_t, const uint64_t*,         \
                            const double*, uint64_t*, int*, double*, const double*);
SL_GP4_DIM_DECL(1) SL_GP4_DIM_DECL(2) SL_GP4_DIM_DECL(3) SL_GP4_DIM_DECL(4)
#if !defined(SL_GP4_DIM) || SL_GP4_DIM == 1
SL_GP4_DIM_ENTRY(1)
#endif
#if !defined(SL_GP4_DIM) || SL_GP4_DIM == 2
SL_GP4_DIM_ENTRY(2)
#endif
#if !defined(SL_GP4_DIM) || SL_GP4_DIM == 3
SL_GP4_DIM_ENTRY(3)
#endif
#if !defined(SL_GP4_DIM) || SL_GP4_DIM == 4
SL_GP4_DIM_ENTRY(4)

// Fast-path models (closed-form or per-vertex table policy, quadratic V) with panels of 512 rows.
int sl_gp4_sweep_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                        const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                        int* nblocks, double* d_dbg, const double* d_points) {
    const int variant = sl_dim_variant_of(model);
#define SL_GP4(D_)                                                                                 \
    return sl_gp4_launch_d##D_(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,     \
                               d_dbg, d_points)
    switch (variant) {
        case 1: SL_GP4(1);
        case 2: SL_GP4(2);
        case 3: SL_GP4(3);
        case 4: SL_GP4(4);
        default: break;
    }
#undef SL_GP4
    return sl_fail(ctx, SL_ERR_UNSUPPORTED, "k_gp_sweep4 is compiled for 1..4 state dimensions and "
                                            "one action dimension");
}

// models this kernel is instantiated for (the others stay on k_gp_sweep)
bool sl_gp4_supports(const SlDevModel& model) {
    return !sl_model_is_general(model) && sl_dim_variant_of(model) >= 1;
}
#endif  // SL_GP4_DIM: host dispatch
#endif  // SL_NO_GP4
