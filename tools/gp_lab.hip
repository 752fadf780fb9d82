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
// V2: v_mfma_f64_4x4x4_4b_f64, one wavefront per SIMD (W = 4, 512 registers), R x CB x 4 rotations
// of accumulators per wavefront.  A fragments = the 16x16x4 layout (block b of the instruction
// = rows 4b..4b+3 of the 16-row block); the four 4-cell groups of a 16-cell block are paired
// with the row groups by reading the k_x fragment rotated by 0/4/8/12 lanes per row of 16.
//   GEN: 0 = no k_x generation (GEMM phase only, garbage results), 1 = exp per (point, cell)
//   DIAG: 0 = the diagonal row block of a chunk runs all 8 slab pairs (zeros above the diagonal),
//         1 = predicated on its count of slab pairs on or below the diagonal
// ---------------------------------------------------------------------------------------------
template <int R, int CB, int SKIP = 0>      // SKIP bit 0: no A loads, bit 1: no B loads (cost attribution)
struct V2 {
    static constexpr int W = 4, C = 16 * CB, RP = 16 * R * W, RB = R * W;
    static constexpr int KXBUF = 16 * CB * 64;
    static_assert(CB == W, "wave w generates cell block w");
    struct BFrag { d2 v[CB]; };                         // one rotation: [cell block]
    struct AFrag { d2 v[R]; };

    static __device__ __forceinline__ void load_b(BFrag& b, const double* kxs, int off) {
        if (SKIP & 2) return;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) b.v[cb] = *reinterpret_cast<const d2*>(kxs + cb * 128 + off);
    }
    // row blocks r >= r0 are active in this chunk: one uniform multiway branch into the unrolled
    // sequence of row blocks (fall-through), no per-block predicates
#define SL_FROM(R0_, BODY)                                                                       \
    switch (R0_) {                                                                               \
        case 0: if (R > 0) { BODY(0) } [[fallthrough]];                                          \
        case 1: if (R > 1) { BODY(1) } [[fallthrough]];                                          \
        case 2: if (R > 2) { BODY(2) } [[fallthrough]];                                          \
        case 3: if (R > 3) { BODY(3) } [[fallthrough]];                                          \
        case 4: if (R > 4) { BODY(4) } [[fallthrough]];                                          \
        case 5: if (R > 5) { BODY(5) } [[fallthrough]];                                          \
        case 6: if (R > 6) { BODY(6) } [[fallthrough]];                                          \
        case 7: if (R > 7) { BODY(7) } [[fallthrough]];                                          \
        default: break;                                                                          \
    }
    static __device__ __forceinline__ void load_a(AFrag& a, const Problem& pr, const int (&rowblk)[R],
                                                  int r0, int s2abs, int lane) {
        if (SKIP & 1) return;
#define SL_LOAD_A(r_)                                                                            \
    {                                                                                            \
        constexpr int r = (r_) < R ? (r_) : 0;                                                   \
        const double* base = pr.mpack + ((size_t)rowblk[r] * pr.nslab2 + (size_t)s2abs) * 128;   \
        a.v[r] = *reinterpret_cast<const d2*>(base + lane * 2);                                  \
    }
        SL_FROM(r0, SL_LOAD_A)
#undef SL_LOAD_A
    }
    template <int ROT>
    static __device__ __forceinline__ void mfmas(double (&acc)[R][CB][4], const AFrag& a,
                                                 const BFrag& b, int r0) {
#define SL_GROUP(r_)                                                                             \
    {                                                                                            \
        constexpr int r = (r_) < R ? (r_) : 0;                                                   \
        _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                        \
            acc[r][cb][ROT] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[r].x, b.v[cb].x,            \
                                                                  acc[r][cb][ROT], 0, 0, 0);     \
        _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                        \
            acc[r][cb][ROT] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[r].y, b.v[cb].y,            \
                                                                  acc[r][cb][ROT], 0, 0, 0);     \
    }
        SL_FROM(r0, SL_GROUP)
#undef SL_GROUP
    }
    // one slab pair: the four rotations, the k_x fragment of the next rotation (or of the next
    // slab pair's first rotation) requested before the MFMAs of the current one
    static __device__ __forceinline__ void slab_pair(double (&acc)[R][CB][4], const AFrag& a,
                                                     BFrag& be, BFrag& bo, const double* kxs,
                                                     const double* kxs_next, int r0,
                                                     const int (&boff)[4]) {
        load_b(bo, kxs, boff[1]);
        mfmas<0>(acc, a, be, r0);
        load_b(be, kxs, boff[2]);
        mfmas<1>(acc, a, bo, r0);
        load_b(bo, kxs, boff[3]);
        mfmas<2>(acc, a, be, r0);
        load_b(be, kxs_next, boff[0]);
        mfmas<3>(acc, a, bo, r0);
    }
    // one chunk of 64 training points: A fragments in ping-pong register sets, those of slab pair
    // s2 + 1 requested before the MFMAs of slab pair s2
    static __device__ __forceinline__ void chunk(double (&acc)[R][CB][4], const Problem& pr,
                                                 const double* kxb, const int (&rowblk)[R],
                                                 int r0, int ch, int lane, const int (&boff)[4]) {
        AFrag a0, a1;
        BFrag be, bo;
        if (SKIP) {
#pragma unroll
            for (int r = 0; r < R; ++r) a0.v[r] = a1.v[r] = (d2){1.0 + lane * 1e-9, 1.0 - lane * 1e-9};
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) be.v[cb] = bo.v[cb] = (d2){1e-3 * lane, 1e-3};
        }
        load_a(a0, pr, rowblk, r0, 8 * ch, lane);
        load_b(be, kxb, boff[0]);
        for (int s2 = 0; s2 < 8; s2 += 2) {
            const double* k0 = kxb + s2 * CB * 128;
            const double* k1 = k0 + CB * 128;
            const double* k2 = (s2 + 2 < 8) ? k1 + CB * 128 : k1;
            load_a(a1, pr, rowblk, r0, 8 * ch + s2 + 1, lane);
            slab_pair(acc, a0, be, bo, k0, k1, r0, boff);
            if (s2 + 2 < 8) load_a(a0, pr, rowblk, r0, 8 * ch + s2 + 2, lane);
            slab_pair(acc, a1, be, bo, k1, k2, r0, boff);
        }
    }
};

template <int R, int CB, int GEN, int SKIP>
__global__ __launch_bounds__(256, 1) void k_v2(const Problem pr, int64_t lo, int64_t ntiles,
                                               double* __restrict__ ss_out,
                                               double* __restrict__ mean_out) {
    using K = V2<R, CB, SKIP>;
    constexpr int W = K::W, C = K::C, RP = K::RP, RB = K::RB, KXBUF = K::KXBUF;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n_pad = pr.n_pad;
    double* xs_l = smem;
    double* alpha_l = xs_l + P * n_pad;
    double* kx_l = alpha_l + n_pad * DOUT;
    double* part_ss = kx_l + 2 * KXBUF;            // [W][4 rot][C]
    double* cell_m = part_ss + W * 4 * C;          // [C][DOUT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 15, lk = lane >> 4, blk = (lane >> 2) & 3, low = lane & 3;
    int boff[4];
#pragma unroll
    for (int rot = 0; rot < 4; ++rot) boff[rot] = 2 * (16 * lk + 4 * ((blk + rot) & 3) + low);
    for (int k = tid; k < P * n_pad; k += W * 64) xs_l[k] = pr.xs[k];
    for (int k = tid; k < n_pad * DOUT; k += W * 64) alpha_l[k] = pr.alpha[k];
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_base = lo + tile * C;
        double xg[P];
        cell_input(pr, tile_base + 16 * wave + lcol, xg);      // this wave generates cell block `wave`
        double ssr[CB][4], gmean[DOUT];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rot = 0; rot < 4; ++rot) ssr[cb][rot] = 0.0;
#pragma unroll
        for (int dd = 0; dd < DOUT; ++dd) gmean[dd] = 0.0;
        const int npanels = n_pad / RP;
        for (int pan = 0; pan < npanels; ++pan) {
            double acc[R][CB][4];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int rot = 0; rot < 4; ++rot) acc[r][cb][rot] = 0.0;
            int rowblk[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int wsel = (r & 1) ? (W - 1 - wave) : wave;     // balance the triangle
                rowblk[r] = pan * RB + r * W + wsel;
            }
            const int nchunks = (pan + 1) * (RP / 64);
            const int first_new_chunk = pan * (RP / 64);
            auto generate = [&](int ch, int buf) {
                if (GEN == 0) return;
                const bool add_mean = ch >= first_new_chunk;
                for (int s = 0; s < 16; ++s) {
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
                    kx_l[buf * KXBUF + ((((s >> 1) * CB + wave) * 64 + lane) << 1) + (s & 1)] = kx;
                }
            };
            generate(0, 0);
            __syncthreads();
            for (int ch = 0; ch < nchunks; ++ch) {
                const int buf = ch & 1;
                const double* kxb = kx_l + buf * KXBUF;
                // row block r of this wave has its diagonal in chunk 8 pan + r: blocks r >= q run
                // all 8 slab pairs of the chunk (the diagonal block's fragments are zero above
                // the diagonal), blocks r < q lie above the diagonal
                const int q = ch - 8 * pan;
                const int r0 = __builtin_amdgcn_readfirstlane(q < 0 ? 0 : q);
                K::chunk(acc, pr, kxb, rowblk, r0, ch, lane, boff);
                if (ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int rot = 0; rot < 4; ++rot)
                        ssr[cb][rot] = fma(acc[r][cb][rot], acc[r][cb][rot], ssr[cb][rot]);
        }
        // rows of a block live in the four lane groups (row = lane >> 4): fold them, then every
        // (wave, rotation) plane holds one partial sum per cell
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
    launch();
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
    Problem pr;
    pr.n = n; pr.n_pad = n_pad; pr.nslab2 = nslab2; pr.variance = 0.03 * 0.03;
    const double kg[4] = {0.9, 2.1, 0.7, 0.4};
    for (int q = 0; q < 4; ++q) pr.kgain[q] = kg[q];
    for (int q = 0; q < P; ++q) pr.inv_ls[q] = 1.0 / 1.5;
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
    pr.xs = d_xs; pr.alpha = d_alpha; pr.mpack = d_mpack; pr.linv = d_linv;

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

    auto report = [&](const char* name, double ms, bool check) {
        double err_ss = 0.0, err_m = 0.0;
        if (check) {
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
        const double cells = (double)ntiles * 64;
        const double tf = flops_cell * cells / (ms * 1e-3) / 1e12;
        const double cyc = ms * 1e-3 * 2.4e9 * ncu * 4 / ((double)ntiles * mfma_tile);
        printf("%-34s %9.3f ms  %6.2f TFLOP/s  %5.1f cyc/mfma4(@2.4GHz)  err ss %.1e mean %.1e %s\n", name, ms,
               tf, cyc, err_ss, err_m, check ? (err_ss < 1e-9 && err_m < 1e-9 ? "OK" : "MISMATCH") : "");
        fflush(stdout);
    };

#define RUN(NAME, KERN, THREADS, LDS, CHECK)                                                        \
    if (!only[0] || strstr(NAME, only)) {                                                           \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN),                                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));            \
        CK(hipMemset(d_ss, 0, (size_t)ntiles * 64 * 8));                                            \
        const int blocks = (int)(ntiles < ncu ? ntiles : ncu);                                      \
        double ms = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(blocks), dim3(THREADS), LDS, 0, pr, \
                                                     lo, ntiles, d_ss, d_mean); }, reps);          \
        CK(hipGetLastError());                                                                      \
        report(NAME, ms, CHECK);                                                                    \
    }

    const size_t lds_common = ((size_t)P * n_pad + (size_t)n_pad * DOUT) * 8;
    const size_t lds_v1 = lds_common + (2 * 16 * 4 * 64 + 8 * 64 + 8 * 16 * DOUT) * 8;
    const size_t lds_v2 = lds_common + (2 * 16 * 4 * 64 + 4 * 4 * 64 + 64 * DOUT) * 8;
    RUN("v1 16x16x4 W8 R4 CB4", (k_v1<8, 4, 4, 1>), 512, lds_v1, true);
    RUN("v1 16x16x4 W8 R4 CB4 nogen", (k_v1<8, 4, 4, 0>), 512, lds_v1, false);
    RUN("v2 4x4x4 R8 CB4", (k_v2<8, 4, 1, 0>), 256, lds_v2, true);
    RUN("v2 4x4x4 R8 CB4 nogen", (k_v2<8, 4, 0, 0>), 256, lds_v2, false);
    RUN("v2 4x4x4 R8 CB4 nogen noA", (k_v2<8, 4, 0, 1>), 256, lds_v2, false);
    RUN("v2 4x4x4 R8 CB4 nogen noB", (k_v2<8, 4, 0, 2>), 256, lds_v2, false);
    RUN("v2 4x4x4 R8 CB4 nogen noA noB", (k_v2<8, 4, 0, 3>), 256, lds_v2, false);
    return 0;
}
