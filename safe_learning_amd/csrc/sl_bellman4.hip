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
int sl_bellman4_policy_launch(sl_ctx*, int64_t, int64_t, double*, double*, int* done) {
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
    int64_t tfrag;              // < 0, or [segments][n_pad / 32][KXBUF]: the last axis' table of a
                                // 32-point chunk in fragment layout (k_bellman4s)
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
    if (pk.tfrag >= 0) {
        // point jj of a chunk, cell c of a segment -> (jj >> 3) KXS2 + (c >> 4) 128 + 32 (jj & 3)
        // + ((jj >> 2) & 1) + 2 (((c & 15) + 4 ((jj & 3) >> 1)) & 15): what the generation of
        // k_bellman4 writes (the padding words were zeroed by the launcher)
        using namespace bm4;
        const int n_last = (int)M.m.grid.num_points[d - 1];
        const int nchunks = n_pad / 32, segs = n_last / C;
        double* tfrag = pack + pk.tfrag;
        for (int64_t t = tid; t < (int64_t)segs * nchunks * 32 * C; t += nthreads) {
            const int c = (int)(t % C);
            const int jj = (int)((t / C) % 32);
            const int ch = (int)((t / (C * 32)) % nchunks);
            const int seg = (int)(t / ((int64_t)C * 32 * nchunks));
            const int j = 32 * ch + jj;
            double x[SL_P];
            sl_index_to_state(M.m.grid, M.gf, d, (int64_t)(seg * C + c), x);
            double v = 0.0;
            if (j < hd.n) {
                const double dlt = hd.xs[(d - 1) * src_pad + j] - x[d - 1] * hd.inv_ls[d - 1];
                v = sl_exp_nonpos(-0.5 * (dlt * dlt));
            }
            const int off = (jj >> 3) * KXS2 + (c >> 4) * 128 + 32 * (jj & 3) + ((jj >> 2) & 1) +
                            2 * (((c & 15) + 4 * ((jj & 3) >> 1)) & 15);
            tfrag[((int64_t)seg * nchunks + ch) * KXBUF + off] = v;
        }
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

// ---------------------------------------------------------------------------------------------
// Policy evaluation, V <- r(x, pi(x)) + gamma V(f(x, pi(x))), for piecewise-constant table policies
// ---------------------------------------------------------------------------------------------
// The greedy policies of the value-iteration loop take their values from a finite action set, and
// the 64 cells of a tile (one segment of a row of the last grid axis) use only a few of them.  The
// means of ONE action for a tile are a quarter block as in k_bellman4 above - its <= 4 rows
// (outputs) given to all four MFMA blocks, each block a different slab - so a tile costs 1024
// MFMAs per distinct action instead of the 9 x 1024 of the max sweep or the 16-column GEMM of
// k_bellman_policy_mfma (sl_bellman.hip, v_mfma_f64_16x16x4_f64, one exponential per slab and lane).
//
// The factors are grouped the other way round than in k_bellman4:
//   mean[dd][cell] = sum_j (A_a[dd][j] P_j) T_last[cell][j],   A_a = sigma^2 E_j(u_a) alpha'[j][dd],
// i.e. the product P_j of the leading axes' tables goes into the A operand (one multiply per lane
// and group of 16 points) and the B operand is the last axis' table alone - the SAME for every tile
// of a segment.  k_bellman4_policy_pack writes it once per sweep in the LDS fragment layout; a
// workgroup copies a 32-point chunk into LDS once for its eight wavefronts (double buffered, one
// barrier per chunk) instead of every wavefront generating its own S chunk from 512 KB of table
// reads per tile (137 GB of L2 traffic per pass at 64^4 x 1024: 19 ms).
//   * k_bellman4_policy_distinct collects the distinct policy values of [lo, hi) (at most PMAX = 64,
//     otherwise the caller keeps k_bellman_policy_mfma); the pack kernel packs A_a per distinct
//     action in quarter-block order;
//   * a workgroup step = eight tiles of one segment, one per wavefront: the tile's distinct actions
//     by ballots, mapped to the launch's list; PG action slots (16 accumulators each) per pass over
//     the chunks; the workgroup makes as many passes as its busiest tile needs;
//   * the epilogue is k_bellman_policy_mfma's: one cell per lane, its own action's mean, prior
//     mean, reward, value-table lookup, residuals.
namespace bm4 {
constexpr int PG = 4;                      // action slots per pass
constexpr int PMAX = 64;                   // distinct (rounded) policy values per launch
constexpr int PROW = 5;                    // staged means of one slot: [32 cells][4 outputs + 1]
constexpr unsigned long long P_EMPTY = 0x7ff8dead0000beefull;   // a NaN no policy produces

struct PolicyPack {
    int64_t btq;                // [n_glob][n_pad / 16][64 lanes]: lane (i, k): row i & 3, point 16 g + 4 (i >> 2) + k
    int64_t tfrag;              // [segments][n_pad / 32][KXBUF]: T_last of a 32-point chunk, fragment layout
    int64_t tab[SL_D];          // T_k [N_k][n_pad], the leading axes
    int32_t n_pad, n_glob;
    unsigned long long action_bits[PMAX];
};

template <int COL0, int H>
__device__ __forceinline__ void stage_slot(const AccQ& q, double* mean_l, int lane) {
    const int g = (lane >> 2) & 3;
#pragma unroll
    for (int cbh = 0; cbh < 2; ++cbh) {
        double sum = q.v[2 * H + cbh][0];
#pragma unroll
        for (int rot = 1; rot < 4; ++rot)
            sum += __shfl(q.v[2 * H + cbh][rot], (lane & ~12) | (((g - rot) & 3) << 2), 64);
        mean_l[(16 * cbh + 4 * g + (lane & 3)) * PROW + COL0 + (lane >> 4)] = sum;
    }
}
// B operands of group sgl (four slabs): all 16 requested at once (with two in flight, as in
// quarter(), a single action slot waits for the LDS latency eight times per chunk)
struct BQ { double v[4][CB]; };
__device__ __forceinline__ void load_bq(BQ& b, const double* kxb, int sgl, const int (&qoff)[4]) {
    const double* base = kxb + 2 * sgl * KXS2;
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) b.v[rot][cb] = base[cb * 128 + qoff[rot]];
}
// the MFMAs of one group for NACT action slots: every B operand feeds NACT of them
template <int NACT>
__device__ __forceinline__ void quarter_n(AccQ (&q)[PG], const double (&aq)[PG], const BQ& b) {
#pragma unroll
    for (int g = 0; g < NACT; ++g) group_q<0>(q[g], aq[g], b.v[0]);
#pragma unroll
    for (int g = 0; g < NACT; ++g) group_q<1>(q[g], aq[g], b.v[1]);
#pragma unroll
    for (int g = 0; g < NACT; ++g) group_q<2>(q[g], aq[g], b.v[2]);
#pragma unroll
    for (int g = 0; g < NACT; ++g) group_q<3>(q[g], aq[g], b.v[3]);
}

// One pass of the workgroup over the chunks: this wavefront's action slots [g0, g0 + NACT) of its
// tile (NACT = 0: it only keeps the chunk copies and barriers going).  Leaves the means of the
// lanes whose action is one of the slots in mean[].
template <int DT, int NACT>
__device__ __forceinline__ void policy_pass(double* chunk_l, double* stage_l,
                                            const double* __restrict__ tfrag,
                                            const double* p_l, int d, int n_pad,
                                            int tid, int lane, const int (&qoff)[4],
                                            const double* btq, const int* slot_action, int g0,
                                            int gid, double* mean_l) {
    const int nchunks = n_pad / 32;
    // A operand of this lane: row dd = lane & 3 of point 16 g + jq of every group g of 16 points
    const int jq = 4 * ((lane >> 2) & 3) + (lane >> 4);
    const double* bq[PG];                              // wave-uniform bases, indexed by the lane
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int a = NACT > 0 ? __builtin_amdgcn_readfirstlane(slot_action[g0 + (g < NACT ? g : 0)]) : 0;
        bq[g] = btq + (int64_t)a * (n_pad / 16) * 64;
    }
    AccQ accq[PG];
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot) accq[g].v[cb][rot] = 0.0;
    // A operands of chunk ch: the packed values of the slots' actions (global, L2) and P_j of the
    // tile's row (LDS, computed once per tile).  Only LOADS here: the products are formed when the
    // chunk is consumed, one iteration later (multiplying right away puts the round trips in
    // front of the MFMAs of the chunk in flight).
    auto load_a = [&](int ch, double (&raw0)[PG], double (&raw1)[PG], double& p0, double& p1) {
        const int j0 = 32 * ch + jq;
        p0 = p_l[j0];
        p1 = p_l[j0 + 16];
#pragma unroll
        for (int g = 0; g < PG; ++g) {
            raw0[g] = g < NACT ? bq[g][(2 * ch) * 64 + lane] : 0.0;
            raw1[g] = g < NACT ? bq[g][(2 * ch + 1) * 64 + lane] : 0.0;
        }
    };
    constexpr int COPIES = (KXBUF + 64 * W - 1) / (64 * W);
    double copy[COPIES];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int r = 0; r < COPIES; ++r) {
            const int e = tid + 64 * W * r;
            copy[r] = e < KXBUF ? tfrag[(int64_t)ch * KXBUF + e] : 0.0;
        }
    };
    auto store_chunk = [&](double* dst) {
#pragma unroll
        for (int r = 0; r < COPIES; ++r) {
            const int e = tid + 64 * W * r;
            if (e < KXBUF) dst[e] = copy[r];
        }
    };
    // Software pipeline: the B chunk ch + 1 is written into the other buffer at the top of
    // iteration ch (its global loads were issued one iteration earlier), the A operands of chunk
    // ch + 1 are requested before the MFMAs of chunk ch.  One barrier per chunk: behind it every
    // wavefront has finished reading `cur` and writing `nxt`.
    load_chunk(0);
    __syncthreads();                                   // the previous pass has left both buffers
    store_chunk(chunk_l);
    load_chunk(nchunks > 1 ? 1 : 0);
    double raw0[PG], raw1[PG], p0 = 1.0, p1 = 1.0;
    if constexpr (NACT > 0) load_a(0, raw0, raw1, p0, p1);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const double* cur = chunk_l + (ch & 1) * KXBUF;
        store_chunk(chunk_l + ((ch + 1) & 1) * KXBUF);
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);
        if constexpr (NACT > 0) {
            double aq0[PG], aq1[PG];
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                aq0[g] = (DT > 1) ? raw0[g] * p0 : raw0[g];
                aq1[g] = (DT > 1) ? raw1[g] * p1 : raw1[g];
            }
            load_a(ch + 1 < nchunks ? ch + 1 : ch, raw0, raw1, p0, p1);
            if constexpr (NACT <= 2) {
                // few MFMAs per B operand: both groups requested up front
                BQ b0, b1;
                load_bq(b0, cur, 0, qoff);
                load_bq(b1, cur, 1, qoff);
                quarter_n<NACT>(accq, aq0, b0);
                quarter_n<NACT>(accq, aq1, b1);
            } else {
                BQ b;
                load_bq(b, cur, 0, qoff);
                quarter_n<NACT>(accq, aq0, b);
                load_bq(b, cur, 1, qoff);
                quarter_n<NACT>(accq, aq1, b);
            }
        }
        __syncthreads();
    }
    if constexpr (NACT > 0) {
#pragma unroll
        for (int g = 0; g < NACT; ++g) retire_q(accq[g]);
        // slot by slot and half by half through the staging buffer: lane (row, block, col) holds
        // partial sums, lane = cell collects its own action's means
#pragma unroll
        for (int g = 0; g < NACT; ++g) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                wave_sync();
                if (h2 == 0) stage_slot<0, 0>(accq[g], stage_l, lane); else stage_slot<0, 1>(accq[g], stage_l, lane);
                wave_sync();
                if (gid == g0 + g && (lane >> 5) == h2) {
#pragma unroll
                    for (int k = 0; k < SL_D; ++k)
                        if (k < d) mean_l[lane * SL_D + k] = stage_l[(lane & 31) * PROW + k];
                }
            }
        }
        wave_sync();
    }
}

}  // namespace bm4

// ---------------------------------------------------------------------------------------------
// k_bellman4s: the GEMM half of the split max sweep with the B operand shared by the workgroup
// ---------------------------------------------------------------------------------------------
// Same rows, accumulators and MFMA groups as k_bellman4, with the factors grouped as in
// k_bellman4_policy: mean[(a, dd)][cell] = sum_j (Bt[(a, dd)][j] P_j) T_last[cell][j].  The product
// P_j of the leading axes' tables (once per tile, in LDS) scales the A fragments after they
// arrive - two multiplies per row block and slab pair instead of one multiply, one L2 load and one
// LDS store per element of S - and the B operand, the last axis' table in fragment layout
// (Pack::tfrag, written once per sweep), is copied into LDS once per workgroup and chunk for all
// eight wavefronts (double buffered, one barrier per chunk).  A fragments of chunk ch + 1 are
// requested as soon as the MFMAs of their slab pair of chunk ch are issued.  Split mode only: the
// means go to means[row][cell] for k_bellman_lookup, one row block at a time through a
// [32 cells][17] staging buffer per wavefront.
namespace bm4 {
constexpr int SROW = 17;

template <int NRB, int R, int H>
__device__ __forceinline__ void stage_row_block(const Acc<NRB>& acc, double* st, int lane) {
    const int b = (lane >> 2) & 3;
#pragma unroll
    for (int cbh = 0; cbh < 2; ++cbh)
#pragma unroll
        for (int rot = 0; rot < 4; ++rot)
            st[(16 * cbh + 4 * ((b + rot) & 3) + (lane & 3)) * SROW + 4 * b + (lane >> 4)] =
                acc.v[R][2 * H + cbh][rot];
}
template <int NRB>
__device__ __forceinline__ void scale_a(AFrag<NRB>& a, double px, double py) {
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
        a.v[r].x = a.v[r].x * px;
        a.v[r].y = a.v[r].y * py;
    }
}
// rows [row0, row0 + nrows) of the staged half to means[row][cell]: lane = (cell, half of the rows)
__device__ __forceinline__ void write_rows(const double* st, int stride_l, int row0, int nrows,
                                           int rows, double* dst, int64_t means_stride, int lane) {
    const int cell = lane & 31;
    for (int row = lane >> 5; row < nrows; row += 2)
        if (row0 + row < rows) dst[(int64_t)(row0 + row) * means_stride + cell] = st[cell * stride_l + row];
}
}  // namespace bm4

template <int DT, int NRB, bool Q>
__global__ __launch_bounds__(64 * bm4::W) void k_bellman4s(
    const SlDevModel M, const SlGpDev gp, bm4::Pack pk, int64_t tfrag_off, int64_t lo, int64_t hi,
    int n_actions, const double* __restrict__ pack, double* __restrict__ means_out,
    int64_t means_stride) {
    using namespace bm4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SlGpHeadDev& hd = gp.head[0];
    const int n_pad = pk.n_pad, nslab2 = pk.nslab2, rows = n_actions * hd.dout;
    const int nchunks = n_pad / 32;
    double* chunk_l = smem;                            // two chunk buffers of the workgroup
    double* stage_l = smem + 2 * KXBUF + (size_t)wave * 32 * SROW;
    double* p_l = smem + 2 * KXBUF + (size_t)W * 32 * SROW + (size_t)wave * n_pad;
    __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(pack + pk.bt), 0, 0x7fffffff, 0x27000);
    const int lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int boff[4], qoff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot) {
        boff[rot] = 2 * (16 * lk + ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15));
        qoff[rot] = (blk >> 1) * KXS2 + 32 * lk + 2 * ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15) + (blk & 1);
    }
    const int jq = 4 * blk + lk;                       // quarter block: point 16 g + jq of group g
    const double* btq = pack + pk.btq;
    constexpr int COPIES = (KXBUF + 64 * W - 1) / (64 * W);

    const int64_t n_last = M.m.grid.num_points[d - 1];
    const int64_t segs = n_last / C;
    const int64_t row_lo = lo / n_last, row_hi = (hi + n_last - 1) / n_last;
    const int64_t nsteps = ((row_hi - row_lo + W - 1) / W) * segs;
    for (int64_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const int64_t seg = step % segs, row = row_lo + (step / segs) * W + wave;
        const int64_t wbase = row * n_last + seg * C;
        const bool live = row < row_hi && wbase >= lo && wbase < hi;   // wave-uniform
        const int64_t tbase = live ? wbase : lo;
        int64_t ijk[SL_D];
        sl_unravel(M.m.grid, M.gf, d, tbase, ijk);
        // P_j = prod_{k < d-1} T_k[i_k][j] of the tile's row, once per tile
        {
            const double* trow[SL_D];
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                const int ik = k < d - 1 ? __builtin_amdgcn_readfirstlane((int)ijk[k]) : 0;
                trow[k] = pack + pk.tab[k < d - 1 ? k : 0] + (int64_t)ik * n_pad;
            }
            for (int j = lane; j < n_pad; j += 64) {
                double v = 1.0;
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < d - 1) v = (k == 0) ? trow[0][j] : v * trow[k][j];
                p_l[j] = v;
            }
        }
        const double* tfrag = pack + tfrag_off + seg * nchunks * (int64_t)KXBUF;
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
        double copy[COPIES];
        auto load_chunk = [&](int ch) {
#pragma unroll
            for (int r = 0; r < COPIES; ++r) {
                const int e = tid + 64 * W * r;
                copy[r] = e < KXBUF ? tfrag[(int64_t)ch * KXBUF + e] : 0.0;
            }
        };
        auto store_chunk = [&](double* dst) {
#pragma unroll
            for (int r = 0; r < COPIES; ++r) {
                const int e = tid + 64 * W * r;
                if (e < KXBUF) dst[e] = copy[r];
            }
        };
        load_chunk(0);
        __syncthreads();                               // the previous step has left both buffers
        store_chunk(chunk_l);
        load_chunk(nchunks > 1 ? 1 : 0);
        AFrag<NRB> a0, a1, a2, a3;
        load_a<NRB>(a0, rsrc, nslab2, 0, lane);
        load_a<NRB>(a1, rsrc, nslab2, 1, lane);
        load_a<NRB>(a2, rsrc, nslab2, 2, lane);
        load_a<NRB>(a3, rsrc, nslab2, 3, lane);
        double aq0 = 0.0, aq1 = 0.0;
        if (Q) {
            aq0 = btq[lane];
            aq1 = btq[64 + lane];
        }
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            const double* cur = chunk_l + (ch & 1) * KXBUF;
            store_chunk(chunk_l + ((ch + 1) & 1) * KXBUF);
            load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);
            const int chn = ch + 1 < nchunks ? ch + 1 : ch;
            // the fragments requested during the previous chunk, scaled by P_j of their points:
            // .x = slab 2 s, .y = slab 2 s + 1 of slab pair s: points 32 ch + 8 s + (0 | 4) + lk
            if (DT > 1) {
                const double* pc = p_l + 32 * ch + lk;
                scale_a<NRB>(a0, pc[0], pc[4]);
                scale_a<NRB>(a1, pc[8], pc[12]);
                scale_a<NRB>(a2, pc[16], pc[20]);
                scale_a<NRB>(a3, pc[24], pc[28]);
                if (Q) {
                    aq0 = aq0 * p_l[32 * ch + jq];
                    aq1 = aq1 * p_l[32 * ch + 16 + jq];
                }
            }
            BFrag be, bo;
            load_b(be, cur, boff[0]);
            slab_pair<NRB>(acc, a0, be, bo, cur, cur + KXS2, boff);
            load_a<NRB>(a0, rsrc, nslab2, 4 * chn, lane);
            slab_pair<NRB>(acc, a1, be, bo, cur + KXS2, cur + 2 * KXS2, boff);
            load_a<NRB>(a1, rsrc, nslab2, 4 * chn + 1, lane);
            if (Q) {
                quarter(accq, aq0, cur, 0, qoff);
                aq0 = btq[(2 * chn) * 64 + lane];
            }
            slab_pair<NRB>(acc, a2, be, bo, cur + 2 * KXS2, cur + 3 * KXS2, boff);
            load_a<NRB>(a2, rsrc, nslab2, 4 * chn + 2, lane);
            slab_pair<NRB>(acc, a3, be, bo, cur + 3 * KXS2, cur + 3 * KXS2, boff);
            load_a<NRB>(a3, rsrc, nslab2, 4 * chn + 3, lane);
            if (Q) {
                quarter(accq, aq1, cur, 1, qoff);
                aq1 = btq[(2 * chn + 1) * 64 + lane];
            }
            __syncthreads();
        }
        retire<NRB>(acc);
        if (Q) retire_q(accq);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            double* dst = means_out + (tbase + SUB * h2 - lo);
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                wave_sync();
                if (h2 == 0) {
                    if (r == 0) stage_row_block<NRB, 0, 0>(acc, stage_l, lane);
                    if (r == 1) stage_row_block<NRB, (NRB > 1 ? 1 : 0), 0>(acc, stage_l, lane);
                    if (r == 2) stage_row_block<NRB, (NRB > 2 ? 2 : 0), 0>(acc, stage_l, lane);
                } else {
                    if (r == 0) stage_row_block<NRB, 0, 1>(acc, stage_l, lane);
                    if (r == 1) stage_row_block<NRB, (NRB > 1 ? 1 : 0), 1>(acc, stage_l, lane);
                    if (r == 2) stage_row_block<NRB, (NRB > 2 ? 2 : 0), 1>(acc, stage_l, lane);
                }
                wave_sync();
                if (live) write_rows(stage_l, SROW, 16 * r, 16, rows, dst, means_stride, lane);
            }
            if (Q) {
                wave_sync();
                if (h2 == 0) stage_slot<0, 0>(accq, stage_l, lane); else stage_slot<0, 1>(accq, stage_l, lane);
                wave_sync();
                if (live) write_rows(stage_l, PROW, 16 * NRB, 4, rows, dst, means_stride, lane);
            }
        }
        wave_sync();
    }
}

// The key the cells of a tile are grouped by, and the action value the packed A operand is built
// from: the policy value rounded to a multiple of 2^-40 (-0 -> +0).  A table policy read at its own
// vertices through the interpolant (SL_POLICY_TRI, what the reference does) returns the vertex
// value only up to the rounding of the barycentric weights (a few 1e-14 on a 64^4 grid);
// k_bellman_policy_mfma groups such values with a 1e-14 tolerance around a leader, here the
// grouping has to be the same in every wavefront of the launch.  The rounding moves an action by
// at most 4.5e-13 (its kernel factor E_j by ~1e-12 relative); the cell's own value is used for
// the prior mean and the reward.
__device__ __forceinline__ unsigned long long sl_b4_action_bits(double u) {
    const double c = fabs(u) < 256.0 ? __builtin_rint(u * 0x1p40) * 0x1p-40 : u;
    return (unsigned long long)__double_as_longlong(c + 0.0);
}

template <int DT>
__global__ __launch_bounds__(256) void k_bellman4_policy_distinct(
    const SlDevModel M, SlAux aux, int64_t lo, int64_t hi, unsigned long long* list,
    double* __restrict__ ucache) {
    using namespace bm4;
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, lane = threadIdx.x & 63;
    // an interpolated policy walks the unit-cell simplices of its table: descriptor in LDS
    __shared__ SlTriLds<true> tri_l;
    aux = sl_stage_aux<true>(tri_l, aux);
    for (int64_t base = lo + (int64_t)blockIdx.x * blockDim.x; base < hi;
         base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = base + threadIdx.x;
        const bool valid = idx < hi;
        const int64_t cidx = valid ? idx : hi - 1;
        double x[SL_P], u[SL_M];
        sl_index_to_state(M.m.grid, M.gf, d, cidx, x);
        sl_policy_any<true>(M, nd, aux.tri, cidx, x, u);
        if (valid) ucache[idx - lo] = u[0];
        const unsigned long long bits = sl_b4_action_bits(u[0]);
        // lanes 0 .. PMAX-1 hold a snapshot of the list: a value that is already in it costs no
        // memory access (2.6e5 wavefronts walking 16 words with compare-and-swap cost 250 ms)
        const unsigned long long seen =
            lane < PMAX ? __atomic_load_n(&list[lane], __ATOMIC_RELAXED) : P_EMPTY;
        uint64_t remaining = __ballot(valid);
        while (remaining) {
            const int leader = __ffsll((unsigned long long)remaining) - 1;
            const unsigned long long lb = __shfl(bits, leader, 64);
            remaining &= ~__ballot(bits == lb);
            if (__ballot(seen == lb) != 0ull) continue;
            if (lane == leader) {
                bool placed = false;
                for (int s = 0; s < PMAX && !placed; ++s) {
                    const unsigned long long old = atomicCAS(&list[s], P_EMPTY, lb);
                    placed = old == P_EMPTY || old == lb;
                }
                if (!placed) atomicExch(&list[PMAX], 1ull);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_bellman4_policy_pack(const SlDevModel M, const SlGpDev gp,
                                                              bm4::PolicyPack pk,
                                                              double* __restrict__ pack) {
    using namespace bm4;
    const int d = M.m.grid.d;
    const SlGpHeadDev& hd = gp.head[0];
    const int src_pad = hd.n_pad, n_pad = pk.n_pad, dout = hd.dout;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    double* btq = pack + pk.btq;
    const int64_t per_action = (int64_t)(n_pad / 16) * 64;
    for (int64_t t = tid; t < per_action * pk.n_glob; t += nthreads) {
        const int a = (int)(t / per_action);
        const int64_t r = t - a * per_action;
        const int l = (int)(r & 63), g = (int)(r >> 6);
        const int dd = l & 3, j = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4);
        double v = 0.0;
        if (dd < dout && j < hd.n) {
            const double ua = __longlong_as_double((long long)pk.action_bits[a]);
            const double dlt = hd.xs[d * src_pad + j] - ua * hd.inv_ls[d];
            v = hd.variance * sl_exp_nonpos(-0.5 * (dlt * dlt)) * hd.alpha[j * dout + dd];
        }
        btq[t] = v;
    }
    int64_t stride = M.m.grid.num_points[d - 1];       // flat-index stride of axis d - 2
    for (int k = d - 2; k >= 0; --k) {
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
    // the last axis in the fragment layout of a chunk buffer (k_bellman4's generation writes):
    // point jj of the chunk, cell c of the segment -> (jj >> 3) KXS2 + (c >> 4) 128 + 32 (jj & 3)
    // + ((jj >> 2) & 1) + 2 (((c & 15) + 4 ((jj & 3) >> 1)) & 15)
    const int n_last = (int)M.m.grid.num_points[d - 1];
    const int nchunks = n_pad / 32, segs = n_last / C;
    double* tfrag = pack + pk.tfrag;
    // (the padding words of the buffers were zeroed by the launcher)
    for (int64_t t = tid; t < (int64_t)segs * nchunks * 32 * C; t += nthreads) {
        const int c = (int)(t % C);
        const int jj = (int)((t / C) % 32);
        const int ch = (int)((t / (C * 32)) % nchunks);
        const int seg = (int)(t / ((int64_t)C * 32 * nchunks));
        const int j = 32 * ch + jj;
        double x[SL_P];
        sl_index_to_state(M.m.grid, M.gf, d, (int64_t)(seg * C + c), x);
        double v = 0.0;
        if (j < hd.n) {
            const double dlt = hd.xs[(d - 1) * src_pad + j] - x[d - 1] * hd.inv_ls[d - 1];
            v = sl_exp_nonpos(-0.5 * (dlt * dlt));
        }
        const int off = (jj >> 3) * KXS2 + (c >> 4) * 128 + 32 * (jj & 3) + ((jj >> 2) & 1) +
                        2 * (((c & 15) + 4 * ((jj & 3) >> 1)) & 15);
        tfrag[((int64_t)seg * nchunks + ch) * KXBUF + off] = v;
    }
}

template <int DT>
__global__ __launch_bounds__(64 * bm4::W) void k_bellman4_policy(
    const SlDevModel M, const SlGpDev gp, SlAux aux, bm4::PolicyPack pk, int64_t lo, int64_t hi,
    const double* __restrict__ pack, const double* __restrict__ ucache, double* __restrict__ v_new,
    double* __restrict__ stats, int flags) {
    // ucache[idx - lo]: the policy's action at every cell (k_bellman4_policy_distinct evaluated it)
    // flags (SL_B4P_FLAGS, diagnostics): 1 no GEMM passes, 2 no value-table lookups
    using namespace bm4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_max[W], red_sum[W];
    __shared__ int slot_action[W][PMAX];
    __shared__ int wave_ng[W];
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* chunk_l = smem;                            // two chunk buffers of the workgroup
    double* stage_l = smem + 2 * KXBUF + (size_t)wave * 32 * PROW;
    double* mean_l = smem + 2 * KXBUF + (size_t)W * 32 * PROW + (size_t)wave * 64 * SL_D;
    double* p_l = smem + 2 * KXBUF + (size_t)W * 32 * PROW + (size_t)W * 64 * SL_D + (size_t)wave * pk.n_pad;
    const int n_pad = pk.n_pad, nchunks = n_pad / 32;
    const int lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int qoff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
        qoff[rot] = (blk >> 1) * KXS2 + 32 * lk + 2 * ((4 * ((blk + rot) & 3) + low + 4 * (lk >> 1)) & 15) + (blk & 1);
    const double* btq = pack + pk.btq;
    int* my_slots = slot_action[wave];

    double lmax = 0.0, lsum = 0.0;
    // workgroup step = eight consecutive rows of the last axis x one 64-cell segment
    const int64_t n_last = M.m.grid.num_points[d - 1];
    const int64_t segs = n_last / C;
    const int64_t row_lo = lo / n_last, row_hi = (hi + n_last - 1) / n_last;
    const int64_t nsteps = ((row_hi - row_lo + W - 1) / W) * segs;
    for (int64_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const int64_t seg = step % segs, row = row_lo + (step / segs) * W + wave;
        const int64_t wbase = row * n_last + seg * C;
        const bool live = row < row_hi && wbase >= lo && wbase < hi;   // wave-uniform
        const int64_t tbase = live ? wbase : lo;          // wave-uniform
        const int64_t idx = tbase + lane;
        int64_t ijk[SL_D];
        sl_unravel(M.m.grid, M.gf, d, tbase, ijk);
        const double* trow[SL_D];                      // wave-uniform row pointers (scalar registers)
#pragma unroll
        for (int k = 0; k < SL_D; ++k) {
            const int ik = k < d - 1 ? __builtin_amdgcn_readfirstlane((int)ijk[k]) : 0;
            trow[k] = pack + pk.tab[k < d - 1 ? k : 0] + (int64_t)ik * n_pad;
        }
        const double u0 = ucache[idx - lo];
        // P_j = prod_{k < d-1} T_k[i_k][j] of the tile's row, once per tile
        for (int j = lane; j < n_pad; j += 64) {
            double v = 1.0;
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d - 1) v = (k == 0) ? trow[0][j] : v * trow[k][j];
            p_l[j] = v;
        }
        // the tile's distinct actions, as indices into the launch's list
        const unsigned long long bits = sl_b4_action_bits(u0);
        int gid = 0, ng = 0;
        uint64_t remaining = live ? ~0ull : 0ull;
        while (remaining) {
            const int leader = __ffsll((unsigned long long)remaining) - 1;
            const unsigned long long lb = __shfl(bits, leader, 64);
            const uint64_t same = __ballot(bits == lb);
            if ((same >> lane) & 1ull) gid = ng;
            int a = 0;
            for (int s = 1; s < pk.n_glob; ++s) a = pk.action_bits[s] == lb ? s : a;
            if (lane == 0) my_slots[ng] = a;
            remaining &= ~same;
            ++ng;
        }
        ng = __builtin_amdgcn_readfirstlane(ng);
        if (lane == 0) wave_ng[wave] = ng;
        __syncthreads();
        int ng_max = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) ng_max = wave_ng[w] > ng_max ? wave_ng[w] : ng_max;
        const double* tfrag = pack + pk.tfrag + seg * nchunks * (int64_t)KXBUF;
#pragma unroll
        for (int k = 0; k < SL_D; ++k) mean_l[lane * SL_D + k] = 0.0;
        for (int g0 = (flags & 1) ? ng_max : 0; g0 < ng_max; g0 += PG) {
            const int nact = ng - g0 < 0 ? 0 : (ng - g0 < PG ? ng - g0 : PG);
            if (nact == 0) policy_pass<DT, 0>(chunk_l, stage_l, tfrag, p_l, d, n_pad, tid, lane, qoff, btq, my_slots, g0, gid, mean_l);
            else if (nact == 1) policy_pass<DT, 1>(chunk_l, stage_l, tfrag, p_l, d, n_pad, tid, lane, qoff, btq, my_slots, g0, gid, mean_l);
            else if (nact == 2) policy_pass<DT, 2>(chunk_l, stage_l, tfrag, p_l, d, n_pad, tid, lane, qoff, btq, my_slots, g0, gid, mean_l);
            else if (nact == 3) policy_pass<DT, 3>(chunk_l, stage_l, tfrag, p_l, d, n_pad, tid, lane, qoff, btq, my_slots, g0, gid, mean_l);
            else policy_pass<DT, 4>(chunk_l, stage_l, tfrag, p_l, d, n_pad, tid, lane, qoff, btq, my_slots, g0, gid, mean_l);
        }
        wave_sync();
        if (live) {
            // (state and action are recomputed here: nothing of the cell stays in registers
            // across the passes)
            double x[SL_P], u[SL_M], prior[SL_D], nxt[SL_D];
            sl_index_to_state(M.m.grid, M.gf, d, idx, x);
            u[0] = ucache[idx - lo];
            sl_append_action(nd, u, x);
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) if (k < d) nxt[k] = mean_l[lane * SL_D + k] + prior[k];
            const double r = sl_quadratic(M.m.reward, p, x);
            double v = (flags & 2) ? nxt[0] : sl_tri_value_fast<DT>(vt, nxt);
            if (M.m.value.negate) v = v * -1.0;
            const double tq = M.m.gamma * v;
            const double q = r + tq;
            v_new[idx - lo] = q;
            double v_old = vt.table[idx * vt.ncols];
            double v_int = (flags & 2) ? x[0] : sl_tri_value_fast<DT>(vt, x);
            if (M.m.value.negate) { v_old = v_old * -1.0; v_int = v_int * -1.0; }
            lmax = fmax(lmax, fabs(q - v_old));
            const double diff = q - v_int;
            lsum = fma(diff, diff, lsum);
        }
        __syncthreads();                               // wave_ng and the slot lists are reused
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

// Sets *done = 1 when k_bellman4_policy took the policy-evaluation sweep: the conditions of the max
// sweep below, a policy with at most PMAX distinct values on [lo, hi).
int sl_bellman4_policy_launch(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_v_new, double* d_stats,
                              int* done) {
    using namespace bm4;
    *done = 0;
    const bool verbose = getenv("SL_BELLMAN4_POLICY_VERBOSE") != nullptr;
#define SL_B4P_DECLINE(why)                                                          \
    do {                                                                             \
        if (verbose) fprintf(stderr, "k_bellman4_policy declined: %s\n", why);      \
        return SL_OK;                                                                \
    } while (0)
    const char* env = getenv("SL_BELLMAN4");
    if (env && env[0] == '0') SL_B4P_DECLINE("SL_BELLMAN4=0");
    env = getenv("SL_BELLMAN4_POLICY");
    if (env && env[0] == '0') SL_B4P_DECLINE("SL_BELLMAN4_POLICY=0");
    const SlDevModel& M = ctx->h_model;
    const int d = M.m.grid.d;
    if (M.m.policy.m != 1 || ctx->h_gp.nheads != 1) SL_B4P_DECLINE("action dimension / number of GP heads");
    const int variant = sl_dim_variant_of(M);
    if (variant != 4 && variant != 2) SL_B4P_DECLINE("state dimension");
    const SlGpHeadHost& hh = ctx->gp_heads[0];
    if (hh.dout != d || hh.col0 != 0) SL_B4P_DECLINE("the head does not cover the state");
    const int n_pad = ((hh.n + 31) / 32) * 32;
    const int64_t n_last = M.m.grid.num_points[d - 1];
    if (n_last % C != 0 || lo % C != 0 || hi % C != 0 || hi <= lo)
        SL_B4P_DECLINE("last axis / index range not a multiple of 64 cells");
    // two chunk buffers of the workgroup, staged means, final means and P_j per wavefront
    const size_t lds = sizeof(double) * (size_t)(2 * KXBUF + W * 32 * PROW + W * 64 * SL_D + W * n_pad);
    if (lds + sizeof(SlTri) + 4096 > 160 * 1024) SL_B4P_DECLINE("too many training points for the LDS budget");
    PolicyPack pk;
    memset(&pk, 0, sizeof(pk));
    pk.n_pad = n_pad;
    int64_t cursor = PMAX + 8;                         // the distinct list sits in front
    pk.btq = cursor;
    cursor += (int64_t)PMAX * (n_pad / 16) * 64;
    pk.tfrag = cursor;
    cursor += (n_last / C) * (int64_t)(n_pad / 32) * KXBUF;
    for (int k = 0; k < d - 1; ++k) {
        pk.tab[k] = cursor;
        cursor += M.m.grid.num_points[k] * (int64_t)n_pad;
    }
    const int64_t ucache_off = cursor;                 // the policy's action at every cell
    cursor += hi - lo;
    const size_t need = sizeof(double) * (size_t)cursor;
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    double* pack = reinterpret_cast<double*>(ctx->d_scratch);
    double* ucache = pack + ucache_off;
    unsigned long long* list = reinterpret_cast<unsigned long long*>(pack);
    unsigned long long h_list[PMAX + 1];
    for (int s = 0; s < PMAX; ++s) h_list[s] = P_EMPTY;
    h_list[PMAX] = 0;
    SL_HIP_CHECK(ctx, hipMemcpyAsync(list, h_list, sizeof(h_list), hipMemcpyHostToDevice, ctx->stream));
    SlAux aux{ctx->d_tri, ctx->d_net};
    {
        const int64_t nblk = (hi - lo + 255) / 256;
        const int blocks = (int)(nblk < 8 * (int64_t)ctx->num_cu ? nblk : 8 * (int64_t)ctx->num_cu);
        if (variant == 4)
            hipLaunchKernelGGL(k_bellman4_policy_distinct<4>, dim3(blocks), dim3(256), 0, ctx->stream,
                               ctx->h_model, aux, lo, hi, list, ucache);
        else
            hipLaunchKernelGGL(k_bellman4_policy_distinct<2>, dim3(blocks), dim3(256), 0, ctx->stream,
                               ctx->h_model, aux, lo, hi, list, ucache);
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    SL_HIP_CHECK(ctx, hipMemcpyAsync(h_list, list, sizeof(h_list), hipMemcpyDeviceToHost, ctx->stream));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_list[PMAX] != 0) SL_B4P_DECLINE("more than 64 distinct policy values (a smooth policy)");
    for (int s = 0; s < PMAX; ++s)
        if (h_list[s] != P_EMPTY) pk.action_bits[pk.n_glob++] = h_list[s];
    if (pk.n_glob == 0) SL_B4P_DECLINE("no policy value found");
#undef SL_B4P_DECLINE
    SL_HIP_CHECK(ctx, hipMemsetAsync(pack + pk.tfrag, 0,
                                     sizeof(double) * (size_t)((n_last / C) * (int64_t)(n_pad / 32) * KXBUF),
                                     ctx->stream));
    hipLaunchKernelGGL(k_bellman4_policy_pack, dim3(512), dim3(256), 0, ctx->stream, ctx->h_model,
                       ctx->h_gp, pk, pack);
    SL_HIP_CHECK(ctx, hipGetLastError());
    const int64_t rows = (hi + n_last - 1) / n_last - lo / n_last;
    const int64_t nsteps = ((rows + W - 1) / W) * (n_last / C);
    const int blocks = (int)(nsteps < ctx->num_cu ? nsteps : ctx->num_cu);
#define SL_B4P_LAUNCH(D_)                                                                         \
    do {                                                                                          \
        auto kern = k_bellman4_policy<D_>;                                                        \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)lds));                                         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, ctx->stream, ctx->h_model,      \
                           ctx->h_gp, aux, pk, lo, hi, pack, ucache, d_v_new, d_stats, flags);    \
    } while (0)
    const char* fenv = getenv("SL_B4P_FLAGS");
    const int flags = fenv ? atoi(fenv) : 0;
    if (variant == 4) SL_B4P_LAUNCH(4); else SL_B4P_LAUNCH(2);
#undef SL_B4P_LAUNCH
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_bellman4_policy<d=%d> (%d distinct actions)", variant == 4 ? 4 : 2,
                   pk.n_glob);
    *done = 1;
    return SL_OK;
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
    // k_bellman4s (B operand shared by the workgroup) for the GEMM half of the split sweep
    const char* shenv = getenv("SL_BELLMAN4_SHARED");
    const size_t lds_s = sizeof(double) * (size_t)(2 * KXBUF + W * 32 * SROW + W * n_pad);
    const bool shared = split && !(shenv && shenv[0] == '0') && lds_s + 2048 <= 160 * 1024;
    pk.tfrag = -1;
    if (shared) {
        pk.tfrag = cursor;
        cursor += (n_last / C) * (int64_t)(n_pad / 32) * KXBUF;
    }
    const size_t need = sizeof(double) * (size_t)cursor;
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    double* pack = reinterpret_cast<double*>(ctx->d_scratch);
    if (shared)
        SL_HIP_CHECK(ctx, hipMemsetAsync(pack + pk.tfrag, 0,
                                         sizeof(double) * (size_t)((n_last / C) * (int64_t)(n_pad / 32) * KXBUF),
                                         ctx->stream));
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
#define SL_B4S_LAUNCH(D_, N_, Q_)                                                                 \
    do {                                                                                          \
        auto kern = k_bellman4s<D_, N_, Q_>;                                                      \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)lds_s));                                       \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds_s, ctx->stream, ctx->h_model,    \
                           ctx->h_gp, pk, pk.tfrag, rlo, rhi, n_actions, pack, means,             \
                           round_cells);                                                          \
    } while (0)
#define SL_B4S_ROWS(D_)                                                   \
    do {                                                                  \
        if (pk.nrb == 1 && pk.quarter) SL_B4S_LAUNCH(D_, 1, true);        \
        else if (pk.nrb == 1) SL_B4S_LAUNCH(D_, 1, false);                \
        else if (pk.nrb == 2 && pk.quarter) SL_B4S_LAUNCH(D_, 2, true);   \
        else if (pk.nrb == 2) SL_B4S_LAUNCH(D_, 2, false);                \
        else SL_B4S_LAUNCH(D_, 3, false);                                 \
    } while (0)
    double* means = split ? pack + means_off : nullptr;
    const int64_t step_cells = split ? round_cells : (hi - lo);
    for (int64_t rlo = lo; rlo < hi; rlo += step_cells) {
        const int64_t rhi = rlo + step_cells < hi ? rlo + step_cells : hi;
        const int64_t wtiles = (rhi - rlo) / C;
        const int64_t wg = (wtiles + W - 1) / W;
        int blocks = (int)(wg < ctx->num_cu ? wg : ctx->num_cu);
        if (shared) {
            const int64_t nrows = (rhi + n_last - 1) / n_last - rlo / n_last;
            const int64_t nsteps = ((nrows + W - 1) / W) * (n_last / C);
            blocks = (int)(nsteps < ctx->num_cu ? nsteps : ctx->num_cu);
            if (variant == 4) SL_B4S_ROWS(4); else SL_B4S_ROWS(2);
        } else {
            if (variant == 4) SL_B4_ROWS(4); else SL_B4_ROWS(2);
        }
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
#undef SL_B4S_ROWS
#undef SL_B4S_LAUNCH
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, shared ? "k_bellman4s<d=%d, row blocks=%d, quarter=%d> + k_bellman_lookup<%d>"
                       : split ? "k_bellman4<d=%d, row blocks=%d, quarter=%d> + k_bellman_lookup<%d>"
                               : "k_bellman4<d=%d, row blocks=%d, quarter=%d> (fused lookup)",
                   variant == 4 ? 4 : 2, pk.nrb, (int)pk.quarter, variant == 4 ? 4 : 2);
    *done = 1;
    return SL_OK;
}
#endif  // SL_NO_BELLMAN4
