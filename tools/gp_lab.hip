// gp_lab.hip - stand-alone laboratory for the GP posterior GEMM of k_gp_sweep (development tool).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gp_lab.hip -o gpurun_out/gp_lab
//   gpurun_out/gp_lab [ntiles] [reps]
//
// No torch, no python: starts in a second on the GPU box, so many kernel variants can be timed
// per gpurun call.  It reproduces the arithmetic core of the sweep on synthetic data -
//     k_x = variance exp(-1/2 |X - x*|^2),  a = Linv k_x (lower triangular),
//     outputs  |a|^2  and  k_x . alpha'  per cell
// - for tiles of 64 consecutive cells of a 128^4-like grid, checks every variant against a naive
// kernel on the first cells and reports ms, TFLOP/s (algorithmic n^2 + n(4p+2) + 2nD + 2n flops per
// cell, as bench.py counts them) and the matrix-pipe cycles per 4x4x4 MFMA equivalent.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(call)                                                                              \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e__), __FILE__, \
                    __LINE__);                                                                \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int P = 5;          // GP inputs (4 states + 1 action)
constexpr int DOUT = 4;
constexpr int GRIDN = 128;

struct Problem {
    int n, n_pad, nslab2;
    const double* xs;         // [P][n_pad]  training inputs / lengthscale
    const double* alpha;      // [n_pad][DOUT]
    const double* mpack;      // MFMA A fragments [row block 16][slab pair 8][lane 64][2]
    const double* mstream;    // the same fragments in the order the V3 wavefronts consume them
    const double* linv;       // dense row-major [n_pad][n_pad] (naive reference only)
    double variance;
    double kgain[4];          // policy u = clamp(k . x, -1, 1)
    double inv_ls[P];
};

// state of cell idx of a 128^4 grid on [-1, 1]^4 plus the saturated linear action, pre-divided
// by the lengthscales
__device__ __forceinline__ void cell_input(const Problem& pr, int64_t idx, double* xg) {
    const double unit = 2.0 / (GRIDN - 1);
    double x[4];
    int64_t t = idx;
#pragma unroll
    for (int q = 3; q >= 0; --q) {
        x[q] = (double)(t % GRIDN) * unit + -1.0;
        t /= GRIDN;
    }
    double u = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) u = u + pr.kgain[q] * x[q];
    u = u < -1.0 ? -1.0 : (u > 1.0 ? 1.0 : u);
#pragma unroll
    for (int q = 0; q < 4; ++q) xg[q] = x[q] * pr.inv_ls[q];
    xg[4] = u * pr.inv_ls[4];
}

// exp(x) for moderate |x|: 13-term Taylor polynomial after range reduction (production's
// sl_exp_nonpos), ~20 instructions, within 2 ulp
__device__ __forceinline__ double fast_exp(double x) {
    x = x < -800.0 ? -800.0 : x;
    const double k = rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;
    q = fma(q, r, 2.08767569878681e-09);
    q = fma(q, r, 2.505210838544172e-08);
    q = fma(q, r, 2.755731922398589e-07);
    q = fma(q, r, 2.7557319223985893e-06);
    q = fma(q, r, 2.48015873015873e-05);
    q = fma(q, r, 1.984126984126984e-04);
    q = fma(q, r, 1.3888888888888889e-03);
    q = fma(q, r, 8.333333333333333e-03);
    q = fma(q, r, 4.1666666666666664e-02);
    q = fma(q, r, 1.6666666666666666e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)k);
}

// a wave-uniform double held in scalar registers
__device__ __forceinline__ double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// naive reference: one workgroup per cell
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_naive(const Problem pr, int64_t base, double* ss_out,
                                               double* mean_out) {
    extern __shared__ double sm[];
    double* kx = sm;                       // [n_pad]
    double* red = sm + pr.n_pad;           // [256]
    const int64_t idx = base + blockIdx.x;
    double xg[P];
    cell_input(pr, idx, xg);
    for (int j = threadIdx.x; j < pr.n_pad; j += 256) {
        double z = 0.0;
        for (int q = 0; q < P; ++q) {
            const double dlt = pr.xs[q * pr.n_pad + j] - xg[q];
            z = fma(dlt, dlt, z);
        }
        kx[j] = j < pr.n ? pr.variance * exp(-0.5 * z) : 0.0;
    }
    __syncthreads();
    double ss = 0.0;
    for (int i = threadIdx.x; i < pr.n; i += 256) {
        double a = 0.0;
        for (int j = 0; j <= i; ++j) a = fma(pr.linv[(size_t)i * pr.n_pad + j], kx[j], a);
        ss = fma(a, a, ss);
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 256; ++k) t += red[k];
        ss_out[blockIdx.x] = t;
    }
    if (threadIdx.x < DOUT) {
        double m = 0.0;
        for (int j = 0; j < pr.n; ++j) m = fma(kx[j], pr.alpha[j * DOUT + threadIdx.x], m);
        mean_out[blockIdx.x * DOUT + threadIdx.x] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// V1: the production structure (v_mfma_f64_16x16x4_f64, W = 8, R = 4, CB = 4), arithmetic core only
// ---------------------------------------------------------------------------------------------
template <int W, int R, int CB, int GEN>
__global__ __launch_bounds__(W * 64) void k_v1(const Problem pr, int64_t lo, int64_t ntiles,
                                               double* __restrict__ ss_out,
                                               double* __restrict__ mean_out) {
    constexpr int C = 16 * CB, RP = 16 * R * W, RB = R * W;
    constexpr int FRAGS = 16 * CB / W, KXBUF = 16 * CB * 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n_pad = pr.n_pad;
    double* xs_l = smem;
    double* alpha_l = xs_l + P * n_pad;
    double* kx_l = alpha_l + n_pad * DOUT;
    double* part_ss = kx_l + 2 * KXBUF;            // [W][C]
    double* part_m = part_ss + W * C;              // [W][16][DOUT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gcb = wave % CB, lcol = lane & 15, lk = lane >> 4;
    for (int k = tid; k < P * n_pad; k += W * 64) xs_l[k] = pr.xs[k];
    for (int k = tid; k < n_pad * DOUT; k += W * 64) alpha_l[k] = pr.alpha[k];
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_base = lo + tile * C;
        double xg[P];
        cell_input(pr, tile_base + 16 * gcb + lcol, xg);
        double ss[CB], gmean[DOUT];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) ss[cb] = 0.0;
#pragma unroll
        for (int dd = 0; dd < DOUT; ++dd) gmean[dd] = 0.0;
        const int npanels = n_pad / RP;
        for (int pan = 0; pan < npanels; ++pan) {
            d4 acc[R][CB];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[r][cb] = (d4){0.0, 0.0, 0.0, 0.0};
            const int nchunks = (pan + 1) * (RP / 64);
            const int first_new_chunk = pan * (RP / 64);
            auto generate = [&](int ch, int buf) {
                if (GEN == 0) return;
                const bool add_mean = ch >= first_new_chunk;
                for (int k = 0; k < FRAGS; ++k) {
                    const int f = wave + k * W;
                    const int s = f / CB;
                    const int j = 64 * ch + 4 * s + lk;
                    double z = 0.0;
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const double dlt = xs_l[q * n_pad + j] - xg[q];
                        z = fma(dlt, dlt, z);
                    }
                    const double kx = pr.variance * exp(-0.5 * z);
                    if (add_mean) {
#pragma unroll
                        for (int dd = 0; dd < DOUT; ++dd) gmean[dd] = fma(kx, alpha_l[j * DOUT + dd], gmean[dd]);
                    }
                    kx_l[buf * KXBUF + ((((s >> 1) * CB + gcb) * 64 + lane) << 1) + (s & 1)] = kx;
                }
            };
            auto load_a = [&](int I, int s2abs) -> d2 {
                const double* base = pr.mpack + ((size_t)I * pr.nslab2 + (size_t)s2abs) * 128;
                return *reinterpret_cast<const d2*>(base + lane * 2);
            };
            generate(0, 0);
            __syncthreads();
            for (int ch = 0; ch < nchunks; ++ch) {
                const int buf = ch & 1;
                int cnt[R], rowblk[R];
                int r0 = R;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int wsel = (r & 1) ? (W - 1 - wave) : wave;
                    rowblk[r] = pan * RB + r * W + wsel;
                    int c = 2 * rowblk[r] + 2 - 8 * ch;
                    c = c > 8 ? 8 : c;
                    cnt[r] = c > 0 ? c : 0;
                    if (cnt[r] > 0 && r0 == R) r0 = r;
                }
                d2 q0 = {0.0, 0.0}, q1 = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r == r0) {
                        q0 = load_a(rowblk[r], 8 * ch);
                        q1 = load_a(rowblk[r], 8 * ch + 1);
                    }
                }
                const bool gen_first = (W < 8) || ((wave & 4) == 0);
                if (gen_first && ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                const double* kxb = kx_l + buf * KXBUF;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int c = cnt[r];
                    for (int s2 = 0; s2 < c; ++s2) {
                        d2 q2 = q1;
                        const int t = s2 + 2;
                        if (t < c) q2 = load_a(rowblk[r], 8 * ch + t);
                        else if (r + 1 < R) q2 = load_a(rowblk[r + 1 < R ? r + 1 : r], 8 * ch + (t - c));
                        d2 b2[CB];
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            b2[cb] = *reinterpret_cast<const d2*>(kxb + (((s2 * CB + cb) * 64 + lane) << 1));
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[r][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0.x, b2[cb].x, acc[r][cb], 0, 0, 0);
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[r][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0.y, b2[cb].y, acc[r][cb], 0, 0, 0);
                        q0 = q1;
                        q1 = q2;
                    }
                }
                if (!gen_first && ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const d4 t = acc[r][cb];
                    ss[cb] = fma(t.x, t.x, ss[cb]);
                    ss[cb] = fma(t.y, t.y, ss[cb]);
                    ss[cb] = fma(t.z, t.z, ss[cb]);
                    ss[cb] = fma(t.w, t.w, ss[cb]);
                }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            ss[cb] += __shfl_xor(ss[cb], 16, 64);
            ss[cb] += __shfl_xor(ss[cb], 32, 64);
        }
#pragma unroll
        for (int dd = 0; dd < DOUT; ++dd) {
            gmean[dd] += __shfl_xor(gmean[dd], 16, 64);
            gmean[dd] += __shfl_xor(gmean[dd], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) part_ss[wave * C + cb * 16 + lane] = ss[cb];
#pragma unroll
            for (int dd = 0; dd < DOUT; ++dd) part_m[(wave * 16 + lane) * DOUT + dd] = gmean[dd];
        }
        __syncthreads();
        if (tid < C) {
            double sumsq = 0.0;
            for (int w = 0; w < W; ++w) sumsq += part_ss[w * C + tid];
            ss_out[tile * C + tid] = sumsq;
            const int cb = tid >> 4, cc = tid & 15;
            for (int dd = 0; dd < DOUT; ++dd) {
                double mu = 0.0;
                for (int w = cb; w < W; w += CB) mu += part_m[(w * 16 + cc) * DOUT + dd];
                mean_out[(tile * C + tid) * DOUT + dd] = mu;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// V3: V2's structure with the accumulators at FIXED accumulator registers a[0:255], touched only
// by inline-asm MFMA groups (no C++ data flow through them): control flow around the groups then
// needs no phi copies, and every MFMA accumulates in place.  acc(r, cb, rot) = a[2 i : 2 i + 1],
// i = (r * CB + cb) * 4 + rot.
// ---------------------------------------------------------------------------------------------
#define SL_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define SL_ALL_AGPRS                                                                               \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", SL_A16(1), SL_A16(2), SL_A16(3),   \
        SL_A16(4), SL_A16(5), SL_A16(6), SL_A16(7), SL_A16(8), SL_A16(9), SL_A16(10), SL_A16(11),   \
        SL_A16(12), SL_A16(13), SL_A16(14), SL_A16(15), SL_A16(16), SL_A16(17), SL_A16(18),        \
        SL_A16(19), SL_A16(20), SL_A16(21), SL_A16(22), SL_A16(23), SL_A16(24), "a250", "a251",     \
        "a252", "a253", "a254", "a255"

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
          "i"(BASE + 12), "i"(BASE + 13), "i"(BASE + 14), "i"(BASE + 15));
}
template <int N>
__device__ __forceinline__ double acc_read() {
    unsigned lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_accvgpr_read_b32 %1, a%c3\n\ts_nop 0"
                 : "=v"(lo), "=v"(hi)
                 : "i"(N), "i"(N + 1));
    return __hiloint2double((int)hi, (int)lo);
}
template <int I, int NACC>
struct AccLoop {
    static __device__ __forceinline__ void zero() {
        acc_zero16<2 * I>();
        if constexpr (2 * I + 16 < 2 * NACC) AccLoop<I + 8, NACC>::zero();
    }
};

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// OPT bit 0: A fragments through buffer loads (scalar offset per fragment, no 64-bit VALU address
// arithmetic); bit 2: the diagonal row block of a chunk runs only its slab pairs on or below the
// diagonal
template <int R, int CB, int SKIP = 0, int OPT = 0>
struct V3 {
    static constexpr int W = 4, C = 16 * CB, RP = 16 * R * W, RB = R * W;
    // k_x chunk in LDS: [slab pair 8][cell block CB][k 4][slot 16][slab of the pair 2]; the slot
    // of cell c16 in row k is c16 ^ 4k and slab pairs are 4 doubles apart modulo the banks, so
    // that both the fragment reads (lane = (k, cell), 16 B) and the generation writes (lane =
    // training point, 8 B, one cell per instruction) are free of bank conflicts
    static constexpr int KXS2 = CB * 128 + 4;
    static constexpr int KXBUF = 8 * KXS2;
    static_assert(CB == 4 && R * CB * 4 * 2 <= 256, "accumulators must fit a[0:255]");
    struct BFrag { d2 v[CB]; };
    struct AFrag { d2 v[R]; };

    static __device__ __forceinline__ void load_b(BFrag& b, const double* kxs, int off) {
        if (SKIP & 2) return;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) b.v[cb] = *reinterpret_cast<const d2*>(kxs + cb * 128 + off);
    }
    // R0 = first active row block of the chunk (static: one straight-line body per value)
    template <int R0>
    static __device__ __forceinline__ void load_a(AFrag& a, const Problem& pr, const int (&rowblk)[R],
                                                  int s2abs, int lane, int abase = 0) {
        if (SKIP & 1) return;
        if (OPT & 1) {
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((OPT & 2) ? pr.mstream : pr.mpack), 0, 0x7fffffff, 0x27000);
#pragma unroll
            for (int r = R0; r < R; ++r) {
                int soff = (rowblk[r] * pr.nslab2 + s2abs) * 1024;
                if (OPT & 2) soff = abase + ((s2abs & 7) * R + r) * 1024;   // stream order
                if (SKIP & 4) soff = r * 1024;                      // L1-resident
                if (SKIP & 8) soff &= (256 * 1024 - 1);             // L2-resident 256 KB window
                a.v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 0));
            }
            return;
        }
#pragma unroll
        for (int r = R0; r < R; ++r) {
            const double* base = pr.mpack + ((size_t)rowblk[r] * pr.nslab2 + (size_t)s2abs) * 128;
            a.v[r] = *reinterpret_cast<const d2*>(base + lane * 2);
        }
    }
    // eight MFMAs of one (row block, rotation): both slabs of the pair for the four cell blocks
    template <int RI, int ROT>
    static __device__ __forceinline__ void group(const d2& av, const BFrag& b) {
        constexpr int N0 = 2 * ((RI * CB + 0) * 4 + ROT), N1 = 2 * ((RI * CB + 1) * 4 + ROT);
        constexpr int N2 = 2 * ((RI * CB + 2) * 4 + ROT), N3 = 2 * ((RI * CB + 3) * 4 + ROT);
        if (OPT & 16) {
            asm volatile(
                "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %0, %2, a[%c10:%c11]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %0, %4, a[%c12:%c13]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %0, %6, a[%c14:%c15]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %0, %8, a[%c16:%c17]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %1, %3, a[%c10:%c11]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %1, %5, a[%c12:%c13]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %1, %7, a[%c14:%c15]\n\t"
                "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %1, %9, a[%c16:%c17]"
                :
                : "v"(av.x), "v"(av.y), "v"(b.v[0].x), "v"(b.v[0].y), "v"(b.v[1].x), "v"(b.v[1].y),
                  "v"(b.v[2].x), "v"(b.v[2].y), "v"(b.v[3].x), "v"(b.v[3].y), "i"(N0), "i"(N0 + 1),
                  "i"(N1), "i"(N1 + 1), "i"(N2), "i"(N2 + 1), "i"(N3), "i"(N3 + 1));
            return;
        }
        asm volatile(
            "s_nop 1\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %0, %2, a[%c10:%c11]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %0, %4, a[%c12:%c13]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %0, %6, a[%c14:%c15]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %0, %8, a[%c16:%c17]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c10:%c11], %1, %3, a[%c10:%c11]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c12:%c13], %1, %5, a[%c12:%c13]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c14:%c15], %1, %7, a[%c14:%c15]\n\t"
            "v_mfma_f64_4x4x4_4b_f64 a[%c16:%c17], %1, %9, a[%c16:%c17]"
            :
            : "v"(av.x), "v"(av.y), "v"(b.v[0].x), "v"(b.v[0].y), "v"(b.v[1].x), "v"(b.v[1].y),
              "v"(b.v[2].x), "v"(b.v[2].y), "v"(b.v[3].x), "v"(b.v[3].y), "i"(N0), "i"(N0 + 1),
              "i"(N1), "i"(N1 + 1), "i"(N2), "i"(N2 + 1), "i"(N3), "i"(N3 + 1));
    }
    // DG: the chunk has a diagonal row block (block R0), active while `diag_on`
    template <int R0, bool DG, int ROT, int RI = R0>
    static __device__ __forceinline__ void mfmas(const AFrag& a, const BFrag& b, bool diag_on) {
        if constexpr (RI < R) {
            if (DG && RI == R0) {
                if (diag_on) group<RI, ROT>(a.v[RI], b);
            } else {
                group<RI, ROT>(a.v[RI], b);
            }
            mfmas<R0, DG, ROT, RI + 1>(a, b, diag_on);
        }
    }
    template <int R0, bool DG>
    static __device__ __forceinline__ void slab_pair(const AFrag& a, BFrag& be, BFrag& bo,
                                                     const double* kxs, const double* kxs_next,
                                                     const int (&boff)[4], bool diag_on) {
        load_b(bo, kxs, boff[1]);
        mfmas<R0, DG, 0>(a, be, diag_on);
        load_b(be, kxs, boff[2]);
        mfmas<R0, DG, 1>(a, bo, diag_on);
        load_b(bo, kxs, boff[3]);
        mfmas<R0, DG, 2>(a, be, diag_on);
        load_b(be, kxs_next, boff[0]);
        mfmas<R0, DG, 3>(a, bo, diag_on);
    }
    template <int R0, bool DG>
    static __device__ __forceinline__ void chunk(const Problem& pr, const double* kxb,
                                                 const int (&rowblk)[R], int ch, int lane,
                                                 const int (&boff)[4], int dc, int abase) {
        if constexpr (R0 < R) {
            AFrag a0, a1;
            BFrag be, bo;
            if (SKIP) {
#pragma unroll
                for (int r = 0; r < R; ++r) a0.v[r] = a1.v[r] = (d2){1.0 + lane * 1e-9, 1.0 - lane * 1e-9};
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) be.v[cb] = bo.v[cb] = (d2){1e-3 * lane, 1e-3};
            }
            if (OPT & 8) {
                // A fragments requested two slab pairs ahead (three register sets)
                AFrag a2;
                load_a<R0>(a0, pr, rowblk, 8 * ch, lane, abase);
                load_a<R0>(a1, pr, rowblk, 8 * ch + 1, lane, abase);
                load_b(be, kxb, boff[0]);
                for (int s2 = 0; s2 < 6; s2 += 3) {
                    const double* k0 = kxb + s2 * KXS2;
                    load_a<R0>(a2, pr, rowblk, 8 * ch + s2 + 2, lane, abase);
                    slab_pair<R0, DG>(a0, be, bo, k0, k0 + KXS2, boff, s2 < dc);
                    load_a<R0>(a0, pr, rowblk, 8 * ch + s2 + 3, lane, abase);
                    slab_pair<R0, DG>(a1, be, bo, k0 + KXS2, k0 + 2 * KXS2, boff, s2 + 1 < dc);
                    load_a<R0>(a1, pr, rowblk, 8 * ch + s2 + 4, lane, abase);
                    slab_pair<R0, DG>(a2, be, bo, k0 + 2 * KXS2, k0 + 3 * KXS2, boff, s2 + 2 < dc);
                }
                slab_pair<R0, DG>(a0, be, bo, kxb + 6 * KXS2, kxb + 7 * KXS2, boff, 6 < dc);
                slab_pair<R0, DG>(a1, be, bo, kxb + 7 * KXS2, kxb + 7 * KXS2, boff, 7 < dc);
                return;
            }
            load_a<R0>(a0, pr, rowblk, 8 * ch, lane, abase);
            load_b(be, kxb, boff[0]);
            for (int s2 = 0; s2 < 8; s2 += 2) {
                const double* k0 = kxb + s2 * KXS2;
                const double* k1 = k0 + KXS2;
                const double* k2 = (s2 + 2 < 8) ? k1 + KXS2 : k1;
                load_a<R0>(a1, pr, rowblk, 8 * ch + s2 + 1, lane, abase);
                slab_pair<R0, DG>(a0, be, bo, k0, k1, boff, s2 < dc);
                if (s2 + 2 < 8) load_a<R0>(a0, pr, rowblk, 8 * ch + s2 + 2, lane, abase);
                slab_pair<R0, DG>(a1, be, bo, k1, k2, boff, s2 + 1 < dc);
            }
        }
    }
    // q = chunk index relative to the panel's diagonal band (q < 0: every block is full)
    static __device__ __forceinline__ void chunk_any(const Problem& pr, const double* kxb,
                                                     const int (&rowblk)[R], int q, int ch, int lane,
                                                     const int (&boff)[4], int wave, int pan) {
        constexpr bool DG = (OPT & 4) != 0;
        // stream order: chunks of panel 0, then of panel 1, ...; per chunk [wave][slab pair][r]
        const int abase = ((4 * pan * (pan + 1) + ch) * W + wave) * (8 * R * 1024);
        int dc = 8;
        if (DG && q >= 0) dc = 2 * ((q & 1) ? (W - 1 - wave) : wave) + 2;
        switch (q) {
            case 0: chunk<0, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 1: chunk<1, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 2: chunk<2, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 3: chunk<3, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 4: chunk<4, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 5: chunk<5, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 6: chunk<6, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            case 7: chunk<7, DG>(pr, kxb, rowblk, ch, lane, boff, dc, abase); break;
            default: chunk<0, false>(pr, kxb, rowblk, ch, lane, boff, 8, abase); break;
        }
    }
    template <int I>
    static __device__ __forceinline__ void squares(double (&ssr)[CB][4]) {
        // I = (r * CB + cb) * 4 + rot
        const double v = acc_read<2 * I>();
        ssr[(I / 4) % CB][I % 4] = fma(v, v, ssr[(I / 4) % CB][I % 4]);
        if constexpr (I + 1 < R * CB * 4) squares<I + 1>(ssr);
    }
};

template <int R, int CB, int GEN, int SKIP, int OPT = 0>
__global__ __launch_bounds__(256, 1) void k_v3(const Problem pr, int64_t lo, int64_t ntiles,
                                               double* __restrict__ ss_out,
                                               double* __restrict__ mean_out) {
    // GEN: 0 none, 1 exp (library) per (cell, point), lane = (k, cell); 3 the same with fast_exp;
    //      2 lane = training point, geometric recurrence along the 16 cells of the wavefront's
    //        cell block where the GP input is affine in the cell index, mean from the LDS copy
    using K = V3<R, CB, SKIP, OPT>;
    constexpr int W = K::W, C = K::C, RP = K::RP, RB = K::RB, KXBUF = K::KXBUF, KXS2 = K::KXS2;
    asm volatile("" ::: SL_ALL_AGPRS);               // the accumulator file belongs to the MFMA groups
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n_pad = pr.n_pad;
    double* xs_l = smem;
    double* alpha_l = xs_l + P * n_pad;
    double* kx_l = alpha_l + n_pad * DOUT;
    double* part_ss = kx_l + 2 * KXBUF;            // [W][4 rot][C]
    double* cell_m = part_ss + W * 4 * C;          // [C][DOUT]
    double* cin = cell_m + C * DOUT;               // [C][P] GP inputs of the tile's cells
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 15, lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int boff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
        boff[rot] = 2 * (16 * lk + ((4 * ((blk + rot) & 3) + low) ^ (4 * lk)));
    const int own = 2 * (16 * lk + (lcol ^ (4 * lk)));            // this lane's own (k, cell) item
    // generation writes of GEN 2: lane = point jj of the chunk -> slab pair jj >> 3, slab (jj >> 2) & 1,
    // row k = jj & 3
    const int wbase = (lane >> 3) * KXS2 + wave * 128 + 32 * (lane & 3) + ((lane >> 2) & 1);
    for (int k = tid; k < P * n_pad; k += W * 64) xs_l[k] = pr.xs[k];
    for (int k = tid; k < n_pad * DOUT; k += W * 64) alpha_l[k] = pr.alpha[k];
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_base = lo + tile * C;
        double xg[P];
        cell_input(pr, tile_base + 16 * wave + lcol, xg);
        // recurrence set-up: is the GP input affine in the cell index over this wave's 16 cells?
        double x0[P], dlt[P], A2 = 0.0, Q = 1.0;
        bool affine = false;
        if (GEN == 2) {
            if (lk == 0) {
#pragma unroll
                for (int q = 0; q < P; ++q) cin[(16 * wave + lcol) * P + q] = xg[q];
            }
            __syncthreads();
            bool ok = true;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                x0[q] = uniform(cin[(16 * wave) * P + q]);
                dlt[q] = uniform(cin[(16 * wave + 1) * P + q] - x0[q]);
                const double end = cin[(16 * wave + 15) * P + q];
                const double pred = fma(15.0, dlt[q], x0[q]);
                ok = ok && fabs(end - pred) <= 1e-13 * fmax(1.0, fabs(end));
                A2 = fma(dlt[q], dlt[q], A2);
            }
            affine = __builtin_amdgcn_readfirstlane((int)ok) != 0;
            A2 = uniform(A2);
            Q = uniform(fast_exp(-A2));
        }
        double ssr[CB][4], gmean[DOUT];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot) ssr[cb][rot] = 0.0;
#pragma unroll
        for (int dd = 0; dd < DOUT; ++dd) gmean[dd] = 0.0;
        const int npanels = n_pad / RP;
        for (int pan = 0; pan < npanels; ++pan) {
            AccLoop<0, R * CB * 4>::zero();
            int rowblk[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int wsel = (r & 1) ? (W - 1 - wave) : wave;
                rowblk[r] = pan * RB + r * W + wsel;
            }
            const int nchunks = (pan + 1) * (RP / 64);
            const int first_new_chunk = pan * (RP / 64);
            auto generate = [&](int ch, int buf) {
                if (GEN == 0) return;
                double* kxw = kx_l + buf * KXBUF;
                if (GEN == 2 && affine) {
                    const int j = 64 * ch + lane;
                    double z = 0.0, bj = 0.0;
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const double d = xs_l[q * n_pad + j] - x0[q];
                        z = fma(d, d, z);
                        bj = fma(d, dlt[q], bj);
                    }
                    double e = pr.variance * fast_exp(-0.5 * z);
                    double rho = fast_exp(bj - 0.5 * A2);
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        kxw[wbase + 2 * (c ^ (4 * (lane & 3)))] = e;
                        e *= rho;
                        rho *= Q;
                    }
                    return;
                }
                if (GEN == 2) {                    // not affine: one exp per (point, cell), lane = point
                    const int j = 64 * ch + lane;
                    for (int c = 0; c < 16; ++c) {
                        double z = 0.0;
#pragma unroll
                        for (int q = 0; q < P; ++q) {
                            const double d = xs_l[q * n_pad + j] - cin[(16 * wave + c) * P + q];
                            z = fma(d, d, z);
                        }
                        kxw[wbase + 2 * (c ^ (4 * (lane & 3)))] = pr.variance * fast_exp(-0.5 * z);
                    }
                    return;
                }
                const bool add_mean = ch >= first_new_chunk;
                for (int s = 0; s < 16; ++s) {
                    const int j = 64 * ch + 4 * s + lk;
                    double z = 0.0;
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const double d = xs_l[q * n_pad + j] - xg[q];
                        z = fma(d, d, z);
                    }
                    const double kx = pr.variance * (GEN == 3 ? fast_exp(-0.5 * z) : exp(-0.5 * z));
                    if (add_mean) {
#pragma unroll
                        for (int dd = 0; dd < DOUT; ++dd) gmean[dd] = fma(kx, alpha_l[j * DOUT + dd], gmean[dd]);
                    }
                    kxw[(s >> 1) * KXS2 + wave * 128 + own + (s & 1)] = kx;
                }
            };
            // GEN 2: posterior mean from the LDS copy of a freshly generated chunk (lane = (k, cell))
            auto mean_pass = [&](int ch, int buf) {
                const double* kxr = kx_l + buf * KXBUF + wave * 128 + own;
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) {
                    const d2 kx = *reinterpret_cast<const d2*>(kxr + s2 * KXS2);
                    const int j = 64 * ch + 8 * s2 + lk;
                    const d2 a0 = *reinterpret_cast<const d2*>(alpha_l + j * DOUT);
                    const d2 a1 = *reinterpret_cast<const d2*>(alpha_l + j * DOUT + 2);
                    const d2 b0 = *reinterpret_cast<const d2*>(alpha_l + (j + 4) * DOUT);
                    const d2 b1 = *reinterpret_cast<const d2*>(alpha_l + (j + 4) * DOUT + 2);
                    gmean[0] = fma(kx.x, a0.x, gmean[0]);
                    gmean[1] = fma(kx.x, a0.y, gmean[1]);
                    gmean[2] = fma(kx.x, a1.x, gmean[2]);
                    gmean[3] = fma(kx.x, a1.y, gmean[3]);
                    gmean[0] = fma(kx.y, b0.x, gmean[0]);
                    gmean[1] = fma(kx.y, b0.y, gmean[1]);
                    gmean[2] = fma(kx.y, b1.x, gmean[2]);
                    gmean[3] = fma(kx.y, b1.y, gmean[3]);
                }
            };
            generate(0, 0);
            __syncthreads();
            for (int ch = 0; ch < nchunks; ++ch) {
                const int buf = ch & 1;
                const double* kxb = kx_l + buf * KXBUF;
                if (GEN == 2 && ch >= first_new_chunk) mean_pass(ch, buf);
                const int q = __builtin_amdgcn_readfirstlane(ch - 8 * pan);
                K::chunk_any(pr, kxb, rowblk, q, ch, lane, boff, wave, pan);
                if (ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                __syncthreads();
            }
            asm volatile("s_nop 15\n\ts_nop 15");       // MFMA results -> accumulator reads
            K::template squares<0>(ssr);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot) {
                ssr[cb][rot] += __shfl_xor(ssr[cb][rot], 16, 64);
                ssr[cb][rot] += __shfl_xor(ssr[cb][rot], 32, 64);
            }
#pragma unroll
        for (int dd = 0; dd < DOUT; ++dd) {
            gmean[dd] += __shfl_xor(gmean[dd], 16, 64);
            gmean[dd] += __shfl_xor(gmean[dd], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rot = 0; rot < 4; ++rot)
                    part_ss[(wave * 4 + rot) * C + 16 * cb + 4 * ((blk + rot) & 3) + low] = ssr[cb][rot];
#pragma unroll
            for (int dd = 0; dd < DOUT; ++dd) cell_m[(16 * wave + lane) * DOUT + dd] = gmean[dd];
        }
        __syncthreads();
        if (tid < C) {
            double sumsq = 0.0;
            for (int k = 0; k < W * 4; ++k) sumsq += part_ss[k * C + tid];
            ss_out[tile * C + tid] = sumsq;
            for (int dd = 0; dd < DOUT; ++dd) mean_out[(tile * C + tid) * DOUT + dd] = cell_m[tid * DOUT + dd];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static double urand(uint64_t& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((s >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
}

struct Timing { double ms; };

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < reps; ++k) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t ntiles = argc > 1 ? atoll(argv[1]) : 16384;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const char* only = argc > 3 ? argv[3] : "";
    const int n = 1024, n_pad = 1024, nslab2 = n_pad / 8;
    const int ncheck = 512;
    uint64_t seed = 12345;
    std::vector<double> xs((size_t)P * n_pad), alpha((size_t)n_pad * DOUT), linv((size_t)n_pad * n_pad, 0.0);
    std::vector<double> mpack((size_t)n_pad * n_pad, 0.0);
    for (auto& v : xs) v = urand(seed) / 1.5;
    for (auto& v : alpha) v = urand(seed);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) linv[(size_t)i * n_pad + j] = urand(seed) * (i == j ? 1.0 : 0.05);
    for (int I = 0; I < n_pad / 16; ++I)
        for (int S2 = 0; S2 < nslab2; ++S2) {
            if (8 * S2 > 16 * I + 15) continue;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int row = 16 * I + (l & 15), col = 4 * (2 * S2 + e) + (l >> 4);
                    mpack[(((size_t)I * nslab2 + S2) * 64 + l) * 2 + e] =
                        (row < n && col <= row) ? linv[(size_t)row * n_pad + col] : 0.0;
                }
        }
    // stream-ordered copy for V3 (W = 4, R = 8): [panel][chunk][wave][slab pair][r][lane][2]
    std::vector<double> mstream;
    {
        const int W = 4, R = 8, RB = 32;
        const int npan = n_pad / (16 * RB);
        size_t total = 0;
        for (int pan = 0; pan < npan; ++pan) total += (size_t)8 * (pan + 1);
        mstream.assign(total * W * 8 * R * 128, 0.0);
        size_t rec = 0;
        for (int pan = 0; pan < npan; ++pan)
            for (int ch = 0; ch < 8 * (pan + 1); ++ch, ++rec)
                for (int w = 0; w < W; ++w)
                    for (int s2 = 0; s2 < 8; ++s2)
                        for (int r = 0; r < R; ++r) {
                            const int wsel = (r & 1) ? (W - 1 - w) : w;
                            const int I = pan * RB + r * W + wsel, S2 = 8 * ch + s2;
                            const double* src = &mpack[((size_t)I * nslab2 + S2) * 128];
                            double* dst = &mstream[(((rec * W + w) * 8 + s2) * R + r) * 128];
                            memcpy(dst, src, 128 * sizeof(double));
                        }
    }
    Problem pr;
    pr.n = n; pr.n_pad = n_pad; pr.nslab2 = nslab2; pr.variance = 0.03 * 0.03;
    const double kg[4] = {0.9, 2.1, 0.7, 0.4};
    for (int q = 0; q < 4; ++q) pr.kgain[q] = kg[q];
    for (int q = 0; q < P; ++q) pr.inv_ls[q] = 1.0 / 1.5;
    double* d_mstream;
    CK(hipMalloc(&d_mstream, mstream.size() * 8));
    CK(hipMemcpy(d_mstream, mstream.data(), mstream.size() * 8, hipMemcpyHostToDevice));
    double *d_xs, *d_alpha, *d_mpack, *d_linv, *d_ss, *d_mean, *d_ss_ref, *d_mean_ref;
    CK(hipMalloc(&d_xs, xs.size() * 8));
    CK(hipMalloc(&d_alpha, alpha.size() * 8));
    CK(hipMalloc(&d_mpack, mpack.size() * 8));
    CK(hipMalloc(&d_linv, linv.size() * 8));
    CK(hipMalloc(&d_ss, (size_t)ntiles * 64 * 8));
    CK(hipMalloc(&d_mean, (size_t)ntiles * 64 * DOUT * 8));
    CK(hipMalloc(&d_ss_ref, ncheck * 8));
    CK(hipMalloc(&d_mean_ref, ncheck * DOUT * 8));
    CK(hipMemcpy(d_xs, xs.data(), xs.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_alpha, alpha.data(), alpha.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mpack, mpack.data(), mpack.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_linv, linv.data(), linv.size() * 8, hipMemcpyHostToDevice));
    pr.xs = d_xs; pr.alpha = d_alpha; pr.mpack = d_mpack; pr.linv = d_linv; pr.mstream = d_mstream;

    const int64_t lo = (int64_t)GRIDN * GRIDN * GRIDN * 60 + 12345 * 64;     // somewhere inside the grid
    hipLaunchKernelGGL(k_naive, dim3(ncheck), dim3(256), (n_pad + 256) * 8, 0, pr, lo, d_ss_ref, d_mean_ref);
    CK(hipDeviceSynchronize());
    std::vector<double> ss_ref(ncheck), mean_ref(ncheck * DOUT);
    CK(hipMemcpy(ss_ref.data(), d_ss_ref, ncheck * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mean_ref.data(), d_mean_ref, ncheck * DOUT * 8, hipMemcpyDeviceToHost));

    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double flops_cell = (double)n * n + (double)n * (4 * P + 2) + 2.0 * n * DOUT + 2.0 * n;
    // 4x4x4-equivalent MFMAs per tile actually needed (16-row x 8-column granularity of the packing)
    double mfma_tile = 0;
    for (int I = 0; I < n_pad / 16; ++I) mfma_tile += (double)((16 * I + 15) / 8 + 1) * 2 * 4 * 4;
    printf("device %s, %d CUs, clock %d MHz; ntiles %lld (%lld cells), reps %d\n", prop.name, ncu,
           prop.clockRate / 1000, (long long)ntiles, (long long)ntiles * 64, reps);

    struct Variant {
        const char* name;
        std::function<void()> launch;
        bool check;
        std::vector<double> ms;
    };
    std::vector<Variant> variants;
    auto report = [&](const Variant& v) {
        double err_ss = 0.0, err_m = 0.0;
        if (v.check) {
            CK(hipMemset(d_ss, 0, (size_t)ntiles * 64 * 8));
            v.launch();
            CK(hipDeviceSynchronize());
            std::vector<double> ss(ncheck), mean(ncheck * DOUT);
            CK(hipMemcpy(ss.data(), d_ss, ncheck * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(mean.data(), d_mean, ncheck * DOUT * 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < ncheck; ++k) {
                err_ss = fmax(err_ss, fabs(ss[k] - ss_ref[k]) / fabs(ss_ref[k]));
                for (int dd = 0; dd < DOUT; ++dd)
                    err_m = fmax(err_m, fabs(mean[k * DOUT + dd] - mean_ref[k * DOUT + dd]) /
                                            (fabs(mean_ref[k * DOUT + dd]) + 1e-12));
            }
        }
        std::vector<double> t = v.ms;
        std::sort(t.begin(), t.end());
        const double ms = t[t.size() / 2], best = t[0];
        const double cells = (double)ntiles * 64;
        const double tf = flops_cell * cells / (ms * 1e-3) / 1e12;
        const double cyc = ms * 1e-3 * 2.4e9 * ncu * 4 / ((double)ntiles * mfma_tile);
        printf("%-34s med %8.3f ms  min %8.3f  %6.2f TFLOP/s  %5.1f cyc/mfma4  err ss %.1e mean %.1e %s\n",
               v.name, ms, best, tf, cyc, err_ss, err_m,
               v.check ? (err_ss < 1e-9 && err_m < 1e-9 ? "OK" : "MISMATCH") : "");
        fflush(stdout);
    };

#define RUN(NAME, KERN, THREADS, LDS, CHECK)                                                        \
    if (!only[0] || strstr(NAME, only)) {                                                           \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN),                                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));            \
        const int blocks = (int)(ntiles < ncu ? ntiles : ncu);                                      \
        const size_t lds_ = (LDS);                                                                  \
        variants.push_back({NAME, [=] { hipLaunchKernelGGL(KERN, dim3(blocks), dim3(THREADS), lds_, \
                                                           0, pr, lo, ntiles, d_ss, d_mean); },     \
                            CHECK, {}});                                                            \
    }

    const size_t lds_common = ((size_t)P * n_pad + (size_t)n_pad * DOUT) * 8;
    const size_t lds_v1 = lds_common + (2 * 16 * 4 * 64 + 8 * 64 + 8 * 16 * DOUT) * 8;
    const size_t lds_v2 = lds_common + (2 * 8 * (4 * 128 + 4) + 4 * 4 * 64 + 64 * DOUT + 64 * P) * 8;
    RUN("v1 16x16x4 W8 R4 CB4", (k_v1<8, 4, 4, 1>), 512, lds_v1, true);
    RUN("v1 16x16x4 W8 R4 CB4 nogen", (k_v1<8, 4, 4, 0>), 512, lds_v1, false);
    RUN("v3 abuf pf2 nogen", (k_v3<8, 4, 0, 0, 9>), 256, lds_v2, false);
    RUN("v3 abuf pf2 gen2", (k_v3<8, 4, 2, 0, 9>), 256, lds_v2, true);
    RUN("v3 abuf pf2 nonop nogen", (k_v3<8, 4, 0, 0, 25>), 256, lds_v2, false);
    RUN("v3 abuf pf2 nonop gen2", (k_v3<8, 4, 2, 0, 25>), 256, lds_v2, true);
    RUN("v3 abuf nogen noA noB", (k_v3<8, 4, 0, 3, 1>), 256, lds_v2, false);
    RUN("v3 abuf nonop nogen noA noB", (k_v3<8, 4, 0, 3, 17>), 256, lds_v2, false);
    // interleaved rounds: every variant once per round, median over the rounds (the chip's
    // clock follows its power budget, so back-to-back repetitions of one variant are biased)
    for (auto& v : variants) { v.launch(); }
    CK(hipDeviceSynchronize());
    for (int round = 0; round < reps; ++round)
        for (auto& v : variants) v.ms.push_back(time_ms(v.launch, 1));
    CK(hipGetLastError());
    for (auto& v : variants) report(v);
    return 0;
}
