// sl_bellman4.hip - the Bellman max sweep of PolicyIteration (reinforcement_learning.py:98-104,
// 213-279) on v_mfma_f64_4x4x4_4b_f64.
//
// Same mathematics as k_bellman_mfma (sl_bellman.hip): with a finite action set the RBF factorises,
// k_x[j] = S_j(x) E_j(u_a), so the posterior means of all (action, output) pairs of a cell are one
// FP64 GEMM  mean[(a, dd)][cell] = sum_j Bt[(a, dd)][j] S[j][cell],  Bt = sigma^2 E_j(u_a) alpha'[j][dd].
// What differs is the instruction: v_mfma_f64_16x16x4_f64 issues every ~100 cycles on gfx950
// (47 TFLOP/s, where k_bellman_mfma's GEMM loop sits), the four-block 4x4x4 one every 16.3
// (76 TFLOP/s), and its 4-row granularity lets the 9 actions x 4 outputs = 36 rows of the cart-pole
// sweep cost 2.25 row blocks of 16 (see "quarter block") where the 16x16x4 kernel pays for 48
// columns.  Structure (fragment layouts of sl_gp4.hip):
//
//  * rows of the GEMM = (action, output) pairs: Bt is packed once per sweep in MFMA A-fragment
//    order [row block][slab pair][lane][2] (k_bellman4_pack) and fetched with buffer loads;
//  * columns = cells.  A wavefront owns 64 consecutive cells of ONE row of the last grid axis
//    (N_last % 64 == 0), so the state factor is S[j][cell] = P_j T_last[i][j] with
//    P_j = prod_{k < d-1} T_k[i_k][j] common to the 64 cells: one multiply per element.  The
//    wavefront generates its own 32-point chunk of S into its private 16.5 KB LDS buffer (lane =
//    training point, layout and rotated fragment reads as in sl_gp4.hip) - no workgroup barrier
//    anywhere: a wavefront's LDS accesses complete in order;
//  * NRB x 4 cell blocks x 4 rotations <= 48 FP64 accumulators per lane.  Unlike k_gp_sweep4's
//    128 they fit beside the operands, and the loop has no data-dependent branches, so they are
//    ordinary values in vector registers (MFMAs in inline asm with read-write operands: the
//    builtin would insist on the accumulator file); two wavefronts per SIMD (256 registers
//    each), so one runs its GEMM while the other generates a chunk or walks the value table
//    (measured at 64^4 x 9 actions x 1024 points with three full row blocks and the fused
//    epilogue: 37.4 ms with two, 48.4 ms with one; the GEMM alone 22.5 ms = 97 % of the
//    instruction's rate; a start-up phase offset between the two changes nothing, prefetching
//    a chunk's table entries across the MFMAs spills);
//  * a last row block with at most 4 rows (36 = 2 x 16 + 4 rows for 9 actions x 4 outputs) would
//    waste three of the four blocks of every instruction.  Its rows are given to ALL four blocks
//    instead, each block taking a different slab of four training points (block b: slab 4 g + b of
//    the group g of four slabs): one instruction then covers 4 rows x 16 cells x 16 points, a
//    quarter of the instructions.  The four blocks' partial sums - over the slab classes, for
//    cell group (b + rot) & 3 - are added across lanes once per tile ("quarter block", Q);
//  * the means go through LDS either to global memory, means[row][cell], for k_bellman_lookup
//    (one thread per cell, three wavefronts per SIMD: prior mean, reward, value-table lookup,
//    first arg-max) - the default, 27.8 ms per sweep - or to the fused epilogue of
//    k_bellman_mfma, one lane per (cell, action group), 28.8 ms (SL_BELLMAN4_SPLIT=0).
#include "sl_common.h"

#ifdef SL_NO_BELLMAN4
// Compiled out: the build's in-place audit of the MFMA loop failed on this toolchain
// (safe_learning_amd/_build.py); the sweeps stay on k_bellman_mfma (sl_bellman.hip).
int sl_bellman4_launch(sl_ctx*, int64_t, int64_t, int, double*, int32_t*, double*, double*, int* done) {
    *done = 0;
    return SL_OK;
}
#else

typedef double sl_d2 __attribute__((ext_vector_type(2)));
typedef unsigned sl_u4 __attribute__((ext_vector_type(4)));

namespace bm4 {

constexpr int CB = 4;                      // cell blocks of 16 per wavefront
constexpr int C = 16 * CB;                 // cells per wavefront tile
constexpr int W = 8;                       // wavefronts per workgroup (two per SIMD)
constexpr int SP = 4;                      // slab pairs per chunk: 32 training points
constexpr int KXS2 = CB * 128 + 4;         // doubles per slab pair (sl_gp4.hip's layout and skew)
constexpr int KXBUF = SP * KXS2;
constexpr int SUB = 32;                    // cells per epilogue step
constexpr int ROWLEN = 49;                 // staged means: [cell][48 columns + 1]
static_assert(SUB * ROWLEN <= KXBUF, "the staged means reuse the wavefront's chunk buffer");

struct BFrag { sl_d2 v[CB]; };
template <int NRB> struct AFrag { sl_d2 v[NRB]; };

template <int NRB> struct Acc { double v[NRB][CB][4]; };      // acc(r, cb, rot)

// Eight MFMAs of one (row block, rotation): both slabs of the pair for the four cell blocks.  A
// dependent FP64 MFMA must not issue right behind its producer (no interlock on gfx950): the
// second use of each accumulator comes three instructions after the first.
template <int NRB, int RI, int ROT>
__device__ __forceinline__ void group(Acc<NRB>& acc, const sl_d2& av, const BFrag& b) {
    asm volatile(
        "v_mfma_f64_4x4x4_4b_f64 %0, %4, %6, %0\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %1, %4, %8, %1\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %2, %4, %10, %2\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %3, %4, %12, %3\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %0, %5, %7, %0\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %1, %5, %9, %1\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %2, %5, %11, %2\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %3, %5, %13, %3"
        : "+v"(acc.v[RI][0][ROT]), "+v"(acc.v[RI][1][ROT]), "+v"(acc.v[RI][2][ROT]),
          "+v"(acc.v[RI][3][ROT])
        : "v"(av.x), "v"(av.y), "v"(b.v[0].x), "v"(b.v[0].y), "v"(b.v[1].x), "v"(b.v[1].y),
          "v"(b.v[2].x), "v"(b.v[2].y), "v"(b.v[3].x), "v"(b.v[3].y));
}
// Quarter block: accq(cb, rot).  One instruction per (group of four slabs, cell block, rotation):
// A = the block's own slab of the <= 4 rows, B = that slab's k_x for cell group (b + rot) & 3.
struct AccQ { double v[CB][4]; };
template <int ROT>
__device__ __forceinline__ void group_q(AccQ& q, double aq, const double (&bq)[CB]) {
    asm volatile(
        "v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %1, %4, %6, %1\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %2, %4, %7, %2\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %3, %4, %8, %3"
        : "+v"(q.v[0][ROT]), "+v"(q.v[1][ROT]), "+v"(q.v[2][ROT]), "+v"(q.v[3][ROT])
        : "v"(aq), "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]));
}
// the four slabs of group sgl (0 / 1) of the chunk: lane (k, b, col) reads point 4 (4 sgl + b) + k
// of cell 16 cb + 4 ((b + rot) & 3) + col; qoff[rot] = its offset inside a (pair, cell block) tile
__device__ __forceinline__ void quarter(AccQ& q, double aq, const double* kxb, int sgl,
                                        const int (&qoff)[4]) {
    double b0[CB], b1[CB];
    const double* base = kxb + 2 * sgl * KXS2;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b0[cb] = base[cb * 128 + qoff[0]];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b1[cb] = base[cb * 128 + qoff[1]];
    group_q<0>(q, aq, b0);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b0[cb] = base[cb * 128 + qoff[2]];
    group_q<1>(q, aq, b1);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b1[cb] = base[cb * 128 + qoff[3]];
    group_q<2>(q, aq, b0);
    group_q<3>(q, aq, b1);
}
__device__ __forceinline__ void retire_q(AccQ& q) {
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(q.v[0][0]), "+v"(q.v[0][1]), "+v"(q.v[0][2]), "+v"(q.v[0][3]),
                   "+v"(q.v[1][0]), "+v"(q.v[1][1]), "+v"(q.v[1][2]), "+v"(q.v[1][3]),
                   "+v"(q.v[2][0]), "+v"(q.v[2][1]), "+v"(q.v[2][2]), "+v"(q.v[2][3]),
                   "+v"(q.v[3][0]), "+v"(q.v[3][1]), "+v"(q.v[3][2]), "+v"(q.v[3][3]));
}
// Partial sums -> means of the quarter block's rows: the value of (row rowi, cell 16 cb + 4 g + col)
// is the sum over rot of accq(cb, rot) in lane (rowi, b = (g - rot) & 3, col); lane (rowi, g, col)
// collects it and stores it behind the NRB full row blocks of half H.
template <int NRB, int H>
__device__ __forceinline__ void stage_quarter(const AccQ& q, double* mean_l, int lane) {
    const int g = (lane >> 2) & 3;
#pragma unroll
    for (int cbh = 0; cbh < 2; ++cbh) {
        double sum = q.v[2 * H + cbh][0];
#pragma unroll
        for (int rot = 1; rot < 4; ++rot)
            sum += __shfl(q.v[2 * H + cbh][rot], (lane & ~12) | (((g - rot) & 3) << 2), 64);
        mean_l[(16 * cbh + 4 * g + (lane & 3)) * ROWLEN + 16 * NRB + (lane >> 4)] = sum;
    }
}

// the last MFMAs have retired before any other instruction reads an accumulator
template <int NRB, int RI = 0>
__device__ __forceinline__ void retire(Acc<NRB>& acc) {
    if constexpr (RI < NRB) {
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc.v[RI][0][0]), "+v"(acc.v[RI][0][1]), "+v"(acc.v[RI][0][2]), "+v"(acc.v[RI][0][3]),
                       "+v"(acc.v[RI][1][0]), "+v"(acc.v[RI][1][1]), "+v"(acc.v[RI][1][2]), "+v"(acc.v[RI][1][3]),
                       "+v"(acc.v[RI][2][0]), "+v"(acc.v[RI][2][1]), "+v"(acc.v[RI][2][2]), "+v"(acc.v[RI][2][3]),
                       "+v"(acc.v[RI][3][0]), "+v"(acc.v[RI][3][1]), "+v"(acc.v[RI][3][2]), "+v"(acc.v[RI][3][3]));
        retire<NRB, RI + 1>(acc);
    }
}
template <int NRB, int ROT, int RI = 0>
__device__ __forceinline__ void mfmas(Acc<NRB>& acc, const AFrag<NRB>& a, const BFrag& b) {
    if constexpr (RI < NRB) {
        group<NRB, RI, ROT>(acc, a.v[RI], b);
        mfmas<NRB, ROT, RI + 1>(acc, a, b);
    }
}
__device__ __forceinline__ void load_b(BFrag& b, const double* kxs, int off) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) b.v[cb] = *reinterpret_cast<const sl_d2*>(kxs + cb * 128 + off);
}
// A fragments (rows = (action, output) pairs) of slab pair s2abs: one coalesced 1 KiB buffer load
// per row block, the fragment's byte offset in the scalar operand
template <int NRB>
__device__ __forceinline__ void load_a(AFrag<NRB>& a, __amdgpu_buffer_rsrc_t rsrc, int nslab2,
                                       int s2abs, int lane) {
#pragma unroll
    for (int r = 0; r < NRB; ++r)
        a.v[r] = __builtin_bit_cast(
            sl_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (r * nslab2 + s2abs) * 1024, 0));
}
// one slab pair: the four rotations, the S fragment of the next rotation (or of the next slab
// pair's first rotation) requested before the MFMAs of the current one
template <int NRB>
__device__ __forceinline__ void slab_pair(Acc<NRB>& acc, const AFrag<NRB>& a, BFrag& be, BFrag& bo,
                                          const double* kxs, const double* kxs_next,
                                          const int (&boff)[4]) {
    load_b(bo, kxs, boff[1]);
    mfmas<NRB, 0>(acc, a, be);
    load_b(be, kxs, boff[2]);
    mfmas<NRB, 1>(acc, a, bo);
    load_b(bo, kxs, boff[3]);
    mfmas<NRB, 2>(acc, a, be);
    load_b(be, kxs_next, boff[0]);
    mfmas<NRB, 3>(acc, a, bo);
}

// wavefront-local ordering of the LDS phases (compiler fence; the hardware keeps a wavefront's
// LDS accesses in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the means of the cells of half H (cell blocks 2H, 2H + 1) into mean_l[cell in half][row]:
// lane (rowi = lane >> 4, b = (lane >> 2) & 3, col = lane & 3) holds row 16 r + 4 b + rowi of
// cell 16 cb + 4 ((b + rot) & 3) + col in acc(r, cb, rot)
template <int NRB, int H>
__device__ __forceinline__ void stage_half(const Acc<NRB>& acc, double* mean_l, int lane) {
    const int b = (lane >> 2) & 3;
#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
        for (int cbh = 0; cbh < 2; ++cbh)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot)
                mean_l[(16 * cbh + 4 * ((b + rot) & 3) + (lane & 3)) * ROWLEN + 16 * r + 4 * b + (lane >> 4)] =
                    acc.v[r][2 * H + cbh][rot];
}

struct Pack {
    int64_t btq;                // quarter block: [n_pad / 16][64 lanes], lane (i, k): row 16 NRB + (i & 3),
                                // point 16 g + 4 (i >> 2) + k
    int64_t bt;                 // Bt fragments [NRB][nslab2][64 lanes][2]
    int64_t tab[SL_D];          // T_k [N_k][n_pad], every axis
    int32_t nslab2, n_pad, nrb, quarter;
};

}  // namespace bm4

// Bt fragments and the per-axis tables of one shared-input GP head.  The training points are
// padded to a multiple of 32 (pk.n_pad, zero rows) independently of the head's own padding.
__global__ __launch_bounds__(256) void k_bellman4_pack(const SlDevModel M, const SlGpDev gp,
                                                       bm4::Pack pk, int n_actions,
                                                       const double* __restrict__ actions,
                                                       double* __restrict__ pack) {
    const int d = M.m.grid.d, m = M.m.policy.m;
    const SlGpHeadDev& hd = gp.head[0];
    const int src_pad = hd.n_pad, n_pad = pk.n_pad, dout = hd.dout;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = (int64_t)pk.nrb * pk.nslab2 * 128;
    double* bt = pack + pk.bt;
    for (int64_t t = tid; t < total; t += nthreads) {
        const int e = (int)(t & 1), l = (int)((t >> 1) & 63);
        const int s2 = (int)((t >> 7) % pk.nslab2), rb = (int)((t >> 7) / pk.nslab2);
        const int col = 16 * rb + (l & 15), j = 8 * s2 + 4 * e + (l >> 4);
        const int a = col / dout, dd = col - a * dout;
        double v = 0.0;
        if (a < n_actions && j < hd.n) {
            double z = 0.0;
            for (int c = 0; c < m; ++c) {
                const double dlt = hd.xs[(d + c) * src_pad + j] - actions[a * m + c] * hd.inv_ls[d + c];
                z = fma(dlt, dlt, z);
            }
            v = hd.variance * sl_exp_nonpos(-0.5 * z) * hd.alpha[j * dout + dd];
        }
        bt[t] = v;
    }
    if (pk.quarter) {
        double* btq = pack + pk.btq;
        for (int64_t t = tid; t < (int64_t)(n_pad / 16) * 64; t += nthreads) {
            const int l = (int)(t & 63), g = (int)(t >> 6);
            const int col = 16 * pk.nrb + (l & 3), j = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4);
            const int a = col / dout, dd = col - a * dout;
            double v = 0.0;
            if (a < n_actions && j < hd.n) {
                double z = 0.0;
                for (int c = 0; c < m; ++c) {
                    const double dlt = hd.xs[(d + c) * src_pad + j] - actions[a * m + c] * hd.inv_ls[d + c];
                    z = fma(dlt, dlt, z);
                }
                v = hd.variance * sl_exp_nonpos(-0.5 * z) * hd.alpha[j * dout + dd];
            }
            btq[t] = v;
        }
    }
    int64_t stride = 1;                    // flat-index stride of axis k (last axis fastest)
    for (int k = d - 1; k >= 0; --k) {
        const int nk = (int)M.m.grid.num_points[k];
        double* tab = pack + pk.tab[k];
        for (int64_t t = tid; t < (int64_t)nk * n_pad; t += nthreads) {
            const int i = (int)(t / n_pad), j = (int)(t % n_pad);
            double x[SL_P];
            sl_index_to_state(M.m.grid, M.gf, d, (int64_t)i * stride, x);
            double v = 0.0;
            if (j < hd.n) {
                const double dlt = hd.xs[k * src_pad + j] - x[k] * hd.inv_ls[k];
                v = sl_exp_nonpos(-0.5 * (dlt * dlt));
            }
            tab[t] = v;
        }
        stride *= nk;
    }
}

template <int DT, int NRB, bool Q>
__global__ __launch_bounds__(64 * bm4::W) void k_bellman4(
    const SlDevModel M, const SlGpDev gp, SlAux aux, bm4::Pack pk, int64_t lo, int64_t hi,
    int n_actions, const double* __restrict__ actions, const double* __restrict__ pack,
    double* __restrict__ v_new, int32_t* __restrict__ argmax, double* __restrict__ q_out,
    double* __restrict__ stats, int flags, double* __restrict__ means_out, int64_t means_stride) {
    // means_out != 0: the kernel stops after the GEMM and writes the posterior means of its cells,
    // means_out[row * means_stride + (cell - lo)]; k_bellman_lookup finishes the sweep.
    // flags (SL_BM_FLAGS, diagnostics): 1 no GEMM, 2 no (cell, action) epilogue, 16 no generation,
    // 64 one working wavefront per SIMD, 128 no value-table lookup
    using namespace bm4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_max[W], red_sum[W];
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p, A = n_actions;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* kxb = smem + (size_t)wave * KXBUF;         // this wavefront's chunk / staged means
    const SlGpHeadDev& hd = gp.head[0];
    const int n_pad = pk.n_pad, nslab2 = pk.nslab2, dout = hd.dout;
    const int nchunks = n_pad / 32;
    __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(pack + pk.bt), 0, 0x7fffffff, 0x27000);

    // fragment reads: lane (k = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3), rotation rot
    const int lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int boff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
        boff[rot] = 2 * (16 * lk + ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15));
    // generation writes: lane = (training point jj = lane & 31 of the chunk, half of the cells):
    // slab pair jj >> 3, slab (jj >> 2) & 1, k = jj & 3; cell c16 of a block sits in slot
    // (c16 + 4 (k >> 1)) & 15
    const int jj = lane & 31, half = lane >> 5;
    const int wswz = 4 * ((jj & 3) >> 1);
    double* w_lo = kxb + (jj >> 3) * KXS2 + (2 * half) * 128 + 32 * (jj & 3) + ((jj >> 2) & 1) + 2 * wswz;
    double* w_hi = w_lo - 8 * wswz;                    // slots that wrap around for wswz = 4
    // quarter block: lane (k, b, col) reads slab 4 sgl + b = pair 2 sgl + (b >> 1), slab b & 1
    int qoff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
        qoff[rot] = (blk >> 1) * KXS2 + 32 * lk + 2 * ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15) + (blk & 1);
    const double* btq = pack + pk.btq;

    double lmax = 0.0, lsum = 0.0;
    const int64_t wtiles = (hi - lo) / C;
    const int nw = (flags & 64) ? W / 2 : W;        // diagnostics: one working wavefront per SIMD
    for (int64_t wt = (wave < nw) ? (int64_t)blockIdx.x * nw + wave : wtiles; wt < wtiles;
         wt += (int64_t)gridDim.x * nw) {
        const int64_t wbase = lo + wt * C;             // 64 cells of one row of the last axis
        int64_t ijk[SL_D];
        sl_unravel(M.m.grid, M.gf, d, wbase, ijk);
        // row pointers of the leading axes' tables and of this segment of the last axis
        const double* trow[SL_D];
#pragma unroll
        for (int k = 0; k < SL_D; ++k)
            if (k < d) trow[k] = pack + pk.tab[k] + (k == d - 1 ? ijk[k] + 32 * half : ijk[k]) * (int64_t)n_pad;
        Acc<NRB> acc;
#pragma unroll
        for (int r = 0; r < NRB; ++r)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rot = 0; rot < 4; ++rot) acc.v[r][cb][rot] = 0.0;
        AccQ accq;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot) accq.v[cb][rot] = 0.0;
        for (int ch = (flags & 1) ? nchunks : 0; ch < nchunks; ++ch) {
            double aq0 = 0.0, aq1 = 0.0;
            if (Q) {
                aq0 = btq[(2 * ch) * 64 + lane];
                aq1 = btq[(2 * ch + 1) * 64 + lane];
            }
            AFrag<NRB> a0, a1, a2, a3;
            load_a<NRB>(a0, rsrc, nslab2, 4 * ch, lane);
            load_a<NRB>(a1, rsrc, nslab2, 4 * ch + 1, lane);
            load_a<NRB>(a2, rsrc, nslab2, 4 * ch + 2, lane);
            load_a<NRB>(a3, rsrc, nslab2, 4 * ch + 3, lane);
            // S[j][cell] = P_j T_last[i][j] for the 32 points of the chunk x this lane's 32 cells
            // (prefetching the next chunk's 32 table entries across the MFMAs was tried: the 64
            // registers spill, 37 -> 48 ms)
            if (!(flags & 16)) {
                const int j = 32 * ch + jj;
                double pj = 1.0;
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < d - 1) pj = (k == 0) ? trow[0][j] : pj * trow[k][j];
                const double* tl = trow[d - 1] + j;
                wave_sync();                           // the previous chunk's fragment reads are issued
                // all 32 entries are requested before the first is used (in batches of 8 the four
                // L2 round trips per chunk cost 0.6 ms per sweep)
#pragma unroll
                for (int c0 = 0; c0 < 32; c0 += 32) {
                    double t[32];
#pragma unroll
                    for (int c = 0; c < 32; ++c) t[c] = tl[(int64_t)(c0 + c) * n_pad];
#pragma unroll
                    for (int c = 0; c < 32; ++c) {
                        const int cc = c0 + c;
                        const double sv = (DT > 1) ? pj * t[c] : t[c];
                        ((cc & 15) < 12 ? w_lo : w_hi)[(cc >> 4) * 128 + 2 * (cc & 15)] = sv;
                    }
                }
            }
            wave_sync();
            BFrag be, bo;
            load_b(be, kxb, boff[0]);
            slab_pair<NRB>(acc, a0, be, bo, kxb, kxb + KXS2, boff);
            slab_pair<NRB>(acc, a1, be, bo, kxb + KXS2, kxb + 2 * KXS2, boff);
            if (Q) quarter(accq, aq0, kxb, 0, qoff);
            slab_pair<NRB>(acc, a2, be, bo, kxb + 2 * KXS2, kxb + 3 * KXS2, boff);
            slab_pair<NRB>(acc, a3, be, bo, kxb + 3 * KXS2, kxb + 3 * KXS2, boff);
            if (Q) quarter(accq, aq1, kxb, 1, qoff);
        }
        retire<NRB>(acc);
        if (Q) retire_q(accq);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            wave_sync();
            if (h2 == 0) stage_half<NRB, 0>(acc, kxb, lane); else stage_half<NRB, 1>(acc, kxb, lane);
            if (Q) { if (h2 == 0) stage_quarter<NRB, 0>(accq, kxb, lane); else stage_quarter<NRB, 1>(accq, kxb, lane); }
            wave_sync();
            const int64_t sbase = wbase + SUB * h2;
            if (means_out) {
                // lane = (cell of the half, half of the rows): 256-byte runs of cells per row
                const int rows = A * dout;
                double* dst = means_out + (sbase - lo) + (lane & (SUB - 1));
                for (int row = lane >> 5; row < rows; row += 2)
                    dst[(int64_t)row * means_stride] = kxb[(lane & (SUB - 1)) * ROWLEN + row];
                continue;
            }
            // ---- (cell, action) pairs: prior mean, reward, value lookup ------------------------
            // lane = (cell of the step, group of actions): the cell's state is computed once,
            // each group walks its share of the actions in ascending order
            constexpr int NG = 64 / SUB;
            const int cell = lane & (SUB - 1), grp = lane / SUB;
            const int apg = (A + NG - 1) / NG;
            const int64_t idx = sbase + cell;
            double x[SL_P], u[SL_M], prior[SL_D], nxt[SL_D];
            sl_index_to_state(M.m.grid, M.gf, d, idx, x);
            double best_q = 0.0;
            int best_a = -1;
            for (int ai = (flags & 2) ? apg : 0; ai < apg; ++ai) {
                const int a = grp * apg + ai;
                if (a < A) {
#pragma unroll
                    for (int c = 0; c < SL_M; ++c) if (c < nd.m) u[c] = actions[a * nd.m + c];
                    sl_append_action(nd, u, x);
                    sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
                    for (int k = 0; k < SL_D; ++k) {
                        if (k < d) {
                            const int dd = k - hd.col0;
                            const double mu = (dd >= 0 && dd < dout) ? kxb[cell * ROWLEN + a * dout + dd] : 0.0;
                            nxt[k] = mu + prior[k];
                        }
                    }
                    const double r = sl_quadratic(M.m.reward, p, x);
                    double v = (flags & 128) ? nxt[0] : sl_tri_value_fast<DT>(vt, nxt);
                    if (M.m.value.negate) v = v * -1.0;
                    const double tq = M.m.gamma * v;
                    const double q = r + tq;
                    if (q_out) q_out[(idx - lo) * A + a] = q;
                    if (best_a < 0 || q > best_q) { best_q = q; best_a = a; }
                }
            }
            // first maximum over the groups in ascending action order
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                const double oq = __shfl(best_q, cell + g * SUB, 64);
                const int oa = __shfl(best_a, cell + g * SUB, 64);
                if (oa >= 0 && (best_a < 0 || oq > best_q)) { best_q = oq; best_a = oa; }
            }
            if (grp == 0) {
                v_new[idx - lo] = best_q;
                if (argmax) argmax[idx - lo] = best_a;
                double v_old = vt.table[idx * vt.ncols];
                if (M.m.value.negate) v_old = v_old * -1.0;
                lmax = fmax(lmax, fabs(best_q - v_old));
            }
        }
        wave_sync();
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
        lsum += __shfl_xor(lsum, o, 64);
    }
    if (lane == 0) { red_max[wave] = lmax; red_sum[wave] = lsum; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < W; ++w) { lmax = fmax(lmax, red_max[w]); lsum += red_sum[w]; }
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        atomicAdd(&stats[1], lsum);
    }
}

// Second half of the split sweep: one thread per cell walks the actions - prior mean, reward,
// value-table lookup, first arg-max - from the means k_bellman4 left in means[row][cell].  The
// lookup is a long chain of dependent FP64 and LDS operations per (cell, action); inside
// k_bellman4 two wavefronts per SIMD (all the register file) cannot hide it (10.3 of 31.8 ms), a
// kernel of its own runs it at four to five wavefronts per SIMD.
#ifndef SL_B4_LOOKUP_BLOCKS
#define SL_B4_LOOKUP_BLOCKS 3
#endif
template <int DT>
__global__ __launch_bounds__(256, SL_B4_LOOKUP_BLOCKS) void k_bellman_lookup(
    const SlDevModel M, const SlGpDev gp, SlAux aux, int64_t lo, int64_t hi, int64_t out_lo,
    int n_actions, const double* __restrict__ actions, const double* __restrict__ means,
    int64_t means_stride, double* __restrict__ v_new, int32_t* __restrict__ argmax,
    double* __restrict__ q_out, double* __restrict__ stats) {
    __shared__ double red_max[4];
    __shared__ SlTri vt_lds;
    __shared__ double act_l[16 * SL_M];
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p, A = n_actions;
    if ((int)threadIdx.x < A * nd.m) act_l[threadIdx.x] = actions[threadIdx.x];
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const SlGpHeadDev& hd = gp.head[0];
    const int dout = hd.dout;
    double lmax = 0.0;
    for (int64_t idx = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * blockDim.x) {
        double x[SL_P], u[SL_M], prior[SL_D], nxt[SL_D];
        sl_index_to_state(M.m.grid, M.gf, d, idx, x);
        const double* mcol = means + (idx - lo);
        double best_q = 0.0;
        int best_a = -1;
        for (int a = 0; a < A; ++a) {
#pragma unroll
            for (int c = 0; c < SL_M; ++c) if (c < nd.m) u[c] = act_l[a * nd.m + c];
            sl_append_action(nd, u, x);
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d) {
                    const int dd = k - hd.col0;
                    const double mu = (dd >= 0 && dd < dout) ? mcol[(int64_t)(a * dout + dd) * means_stride] : 0.0;
                    nxt[k] = mu + prior[k];
                }
            }
            const double r = sl_quadratic(M.m.reward, p, x);
            double v = sl_tri_value_fast<DT>(vt, nxt);
            if (M.m.value.negate) v = v * -1.0;
            const double tq = M.m.gamma * v;
            const double q = r + tq;
            if (q_out) q_out[(idx - out_lo) * A + a] = q;
            if (best_a < 0 || q > best_q) { best_q = q; best_a = a; }
        }
        v_new[idx - out_lo] = best_q;
        if (argmax) argmax[idx - out_lo] = best_a;
        double v_old = vt.table[idx * vt.ncols];
        if (M.m.value.negate) v_old = v_old * -1.0;
        lmax = fmax(lmax, fabs(best_q - v_old));
    }
    for (int o = 32; o >= 1; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
    if ((threadIdx.x & 63) == 0) red_max[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) lmax = fmax(lmax, red_max[w]);
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
    }
}

// Sets *done = 1 when this kernel took the sweep: one shared-input GP head covering the state,
// 2 or 4 state dimensions, one action dimension, at most 48 (action, output) rows, the last grid
// axis a multiple of 64 cells and a 64-aligned index range.
int sl_bellman4_launch(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, double* d_v_new,
                       int32_t* d_argmax, double* d_q, double* d_stats, int* done) {
    using namespace bm4;
    *done = 0;
    const char* env = getenv("SL_BELLMAN4");
    if (env && env[0] == '0') return SL_OK;
    const SlDevModel& M = ctx->h_model;
    const int d = M.m.grid.d;
    if (n_actions < 1 || M.m.policy.m != 1 || ctx->h_gp.nheads != 1) return SL_OK;
    const int variant = sl_dim_variant_of(M);
    if (variant != 4 && variant != 2) return SL_OK;
    const SlGpHeadHost& hh = ctx->gp_heads[0];
    if (hh.dout != d || hh.col0 != 0) return SL_OK;
    const int rows = n_actions * hh.dout;
    if (rows > 48) return SL_OK;
    const int n_pad = ((hh.n + 31) / 32) * 32;
    const int64_t n_last = M.m.grid.num_points[d - 1];
    if (n_last % C != 0 || lo % C != 0 || hi % C != 0 || hi <= lo) return SL_OK;
    Pack pk;
    memset(&pk, 0, sizeof(pk));
    // full row blocks, and a quarter block for a remainder of at most 4 rows
    pk.quarter = (rows > 16 && rows % 16 >= 1 && rows % 16 <= 4) ? 1 : 0;
    pk.nrb = pk.quarter ? rows / 16 : (rows + 15) / 16;
    {
        const char* qenv = getenv("SL_BELLMAN4_QUARTER");
        if (qenv && qenv[0] == '0' && pk.quarter) { pk.quarter = 0; pk.nrb = (rows + 15) / 16; }
    }
    pk.n_pad = n_pad;
    pk.nslab2 = n_pad / 8;
    int64_t cursor = 0;
    pk.btq = cursor;
    cursor += (int64_t)(n_pad / 16) * 64;
    pk.bt = cursor;
    cursor += (int64_t)pk.nrb * pk.nslab2 * 128;
    for (int k = 0; k < d; ++k) {
        pk.tab[k] = cursor;
        cursor += M.m.grid.num_points[k] * (int64_t)n_pad;
    }
    if ((int64_t)pk.nrb * pk.nslab2 * 1024 > 0x7fffffffll) return SL_OK;
    // split sweep (default): the GEMM kernel leaves the means of a round of cells in the scratch
    // buffer and k_bellman_lookup finishes them; SL_BELLMAN4_SPLIT=0 keeps the fused epilogue
    const char* senv = getenv("SL_BELLMAN4_SPLIT");
    const bool split = !(senv && senv[0] == '0');
    const int64_t round_cells = split ? ((hi - lo) < (4ll << 20) ? (hi - lo) : (4ll << 20)) : 0;
    const int64_t means_off = cursor;
    cursor += (int64_t)rows * round_cells;
    const size_t need = sizeof(double) * (size_t)cursor;
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    double* pack = reinterpret_cast<double*>(ctx->d_scratch);
    hipLaunchKernelGGL(k_bellman4_pack, dim3(512), dim3(256), 0, ctx->stream, ctx->h_model, ctx->h_gp,
                       pk, n_actions, ctx->d_actions, pack);
    SL_HIP_CHECK(ctx, hipGetLastError());
    const size_t lds = sizeof(double) * (size_t)W * KXBUF;
    SlAux aux{ctx->d_tri, ctx->d_net};
    const char* fenv = getenv("SL_BM_FLAGS");
    const int flags = fenv ? atoi(fenv) : 0;
#define SL_B4_LAUNCH(D_, N_, Q_)                                                                  \
    do {                                                                                          \
        auto kern = k_bellman4<D_, N_, Q_>;                                                       \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)lds));                                         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, ctx->stream, ctx->h_model,      \
                           ctx->h_gp, aux, pk, rlo, rhi, n_actions, ctx->d_actions, pack,         \
                           d_v_new ? d_v_new + (rlo - lo) : d_v_new,                              \
                           d_argmax ? d_argmax + (rlo - lo) : d_argmax,                           \
                           d_q ? d_q + (rlo - lo) * n_actions : d_q, d_stats, flags, means,       \
                           round_cells);                                                          \
    } while (0)
#define SL_B4_ROWS(D_)                                                    \
    do {                                                                  \
        if (pk.nrb == 1 && pk.quarter) SL_B4_LAUNCH(D_, 1, true);         \
        else if (pk.nrb == 1) SL_B4_LAUNCH(D_, 1, false);                 \
        else if (pk.nrb == 2 && pk.quarter) SL_B4_LAUNCH(D_, 2, true);    \
        else if (pk.nrb == 2) SL_B4_LAUNCH(D_, 2, false);                 \
        else SL_B4_LAUNCH(D_, 3, false);                                  \
    } while (0)
    double* means = split ? pack + means_off : nullptr;
    const int64_t step_cells = split ? round_cells : (hi - lo);
    for (int64_t rlo = lo; rlo < hi; rlo += step_cells) {
        const int64_t rhi = rlo + step_cells < hi ? rlo + step_cells : hi;
        const int64_t wtiles = (rhi - rlo) / C;
        const int64_t wg = (wtiles + W - 1) / W;
        const int blocks = (int)(wg < ctx->num_cu ? wg : ctx->num_cu);
        if (variant == 4) SL_B4_ROWS(4); else SL_B4_ROWS(2);
        SL_HIP_CHECK(ctx, hipGetLastError());
        if (split) {
            const int64_t nblk = (rhi - rlo + 255) / 256;
            const int lblocks = (int)(nblk < 16 * (int64_t)ctx->num_cu ? nblk : 16 * (int64_t)ctx->num_cu);
            if (variant == 4)
                hipLaunchKernelGGL(k_bellman_lookup<4>, dim3(lblocks), dim3(256), 0, ctx->stream,
                                   ctx->h_model, ctx->h_gp, aux, rlo, rhi, lo, n_actions, ctx->d_actions,
                                   means, round_cells, d_v_new, d_argmax, d_q, d_stats);
            else
                hipLaunchKernelGGL(k_bellman_lookup<2>, dim3(lblocks), dim3(256), 0, ctx->stream,
                                   ctx->h_model, ctx->h_gp, aux, rlo, rhi, lo, n_actions, ctx->d_actions,
                                   means, round_cells, d_v_new, d_argmax, d_q, d_stats);
            SL_HIP_CHECK(ctx, hipGetLastError());
        }
    }
#undef SL_B4_ROWS
#undef SL_B4_LAUNCH
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, split ? "k_bellman4<d=%d, row blocks=%d, quarter=%d> + k_bellman_lookup<%d>"
                                     : "k_bellman4<d=%d, row blocks=%d, quarter=%d> (fused lookup)",
                   variant == 4 ? 4 : 2, pk.nrb, (int)pk.quarter, variant == 4 ? 4 : 2);
    *done = 1;
    return SL_OK;
}
#endif  // SL_NO_BELLMAN4
