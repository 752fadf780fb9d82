// sl_nn.hip - Lyapunov check and value pass for a LyapunovNetwork value function
// (examples/utilities.py:85-104; config C3 of BASELINE.json).
//
// Both kernels run the network on the FP64 matrix cores (see nn_mfma_eval).  The posterior (mean,
// error) of GP dynamics comes from a first pass of k_gp_sweep that only emits its per-cell
// records; deterministic dynamics are evaluated here.
#include "sl_common.h"

__device__ __forceinline__ void nn_lv_from_grad(int kind, int d, const double* g, double* lv) {
    if (kind == SL_LIP_ABS_GRAD) {
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) lv[k] = fabs(g[k]);
    } else {
        double acc = fabs(g[0]);
#pragma unroll
        for (int k = 1; k < SL_D; ++k) if (k < d) acc = acc + fabs(g[k]);
        lv[0] = acc;
    }
}

// A wavefront owns 16 cells at a time.  A layer is the FP64 MFMA GEMM  H_out[out x 16 cells] =
// W[out x in] . H_in[in x 16 cells] on v_mfma_f64_16x16x4_f64: A fragments (lane (i, k) =
// W[16 fb + i][4 s + k]) come from a zero-padded row-major copy of the layer kernels in LDS, and
// the accumulator layout of the instruction - lane (n, g) holds feature 16 fb + 4 r + g of cell n
// in register r - is exactly the B-fragment layout of slab 4 fb + r of the next layer, so the
// activations of all layers stay in registers and the layers chain without any data movement.
// The input gradient is the transposed chain (A fragments W[4 s + k][16 ib + i] from the same LDS
// copy), V = sum of squares folded over the four lane groups with two shuffles.
typedef double sl_nd4 __attribute__((ext_vector_type(4)));
#define SL_NNM_WAVES 8

__device__ __forceinline__ double nd4_get(const sl_nd4& v, int r) {
    return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w));
}

// z: the input of this lane's cell (lane & 15), identical in the four lane groups.  Returns V in
// every lane; grad[k] (k < d) in every lane when want_grad.
template <int NL>
__device__ __forceinline__ double nn_mfma_eval(const SlNet& net, const double* __restrict__ wl,
                                               int lane, const double* z, int d, bool want_grad,
                                               double* grad) {
    const int li = lane & 15, lg = lane >> 4;
    // input as B fragments: slab s holds feature 4 s + lg
    double xin[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * s + q < SL_D) v = (lg == q && 4 * s + q < d) ? z[4 * s + q] : v;
        xin[s] = v;
    }
    sl_nd4 h[NL][4];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int nslab = net.nslab[l], nfb = net.nfb[l], stride = net.wstride[l], a = net.act[l];
        const double* __restrict__ W = wl + net.woff[l] + li * stride + lg;
        sl_nd4 acc[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) acc[fb] = (sl_nd4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (l == 0 && s >= 2) break;
            if (s < nslab) {
                const double b = (l == 0) ? xin[s & 1] : nd4_get(h[l > 0 ? l - 1 : 0][s >> 2], s & 3);
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
                    if (fb < nfb)
                        acc[fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(W[16 * fb * stride + 4 * s], b,
                                                                       acc[fb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            h[l][fb].x = sl_act(a, acc[fb].x);
            h[l][fb].y = sl_act(a, acc[fb].y);
            h[l][fb].z = sl_act(a, acc[fb].z);
            h[l][fb].w = sl_act(a, acc[fb].w);
        }
    }
    double value = 0.0;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        value = fma(h[NL - 1][fb].x, h[NL - 1][fb].x, value);
        value = fma(h[NL - 1][fb].y, h[NL - 1][fb].y, value);
        value = fma(h[NL - 1][fb].z, h[NL - 1][fb].z, value);
        value = fma(h[NL - 1][fb].w, h[NL - 1][fb].w, value);
    }
    value += __shfl_xor(value, 16, 64);
    value += __shfl_xor(value, 32, 64);
    if (!want_grad) return value;
    // backward: t = dV/d(pre-activation) of the current layer, in B-fragment layout
    sl_nd4 t[4];
    {
        const int a = net.act[NL - 1];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const sl_nd4 hh = h[NL - 1][fb];
            t[fb].x = 2.0 * hh.x * sl_dact(a, hh.x > 0.0 ? 1.0 : -1.0, hh.x);
            t[fb].y = 2.0 * hh.y * sl_dact(a, hh.y > 0.0 ? 1.0 : -1.0, hh.y);
            t[fb].z = 2.0 * hh.z * sl_dact(a, hh.z > 0.0 ? 1.0 : -1.0, hh.z);
            t[fb].w = 2.0 * hh.w * sl_dact(a, hh.w > 0.0 ? 1.0 : -1.0, hh.w);
        }
    }
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
        const int nfb = net.nfb[l], nib = net.nib[l], stride = net.wstride[l];
        const double* __restrict__ W = wl + net.woff[l] + lg * stride + li;
        sl_nd4 acc[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[ib] = (sl_nd4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if ((s >> 2) < nfb) {
                const double b = nd4_get(t[s >> 2], s & 3);
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
                    if (ib < nib && (l > 0 || ib == 0))
                        acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(W[4 * s * stride + 16 * ib], b,
                                                                       acc[ib], 0, 0, 0);
            }
        }
        if (l > 0) {
            const int a = net.act[l - 1];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const sl_nd4 hh = h[l > 0 ? l - 1 : 0][ib];
                t[ib].x = acc[ib].x * sl_dact(a, hh.x > 0.0 ? 1.0 : -1.0, hh.x);
                t[ib].y = acc[ib].y * sl_dact(a, hh.y > 0.0 ? 1.0 : -1.0, hh.y);
                t[ib].z = acc[ib].z * sl_dact(a, hh.z > 0.0 ? 1.0 : -1.0, hh.z);
                t[ib].w = acc[ib].w * sl_dact(a, hh.w > 0.0 ? 1.0 : -1.0, hh.w);
            }
        } else {
            // input feature k = 4 r + g sits in register r of lane group g
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d) grad[k] = __shfl(nd4_get(acc[0], k >> 2), li + 16 * (k & 3), 64);
        }
    }
    return value;
}

__device__ __forceinline__ void nn_stage_weights(const SlNet& net, double* wl) {
    for (int k = threadIdx.x; k < net.wtotal; k += blockDim.x) wl[k] = net.wpad[k];
    __syncthreads();
}

template <int NL>
__global__ __launch_bounds__(64 * SL_NNM_WAVES) void k_nn_values_mfma(const SlDevModel M, SlAux aux,
                                                                      int64_t lo, int64_t hi,
                                                                      double* __restrict__ values) {
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const SlNet& net = *aux.net;
    nn_stage_weights(net, wl);
    const int d = M.m.grid.d, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t step = (int64_t)gridDim.x * SL_NNM_WAVES * 16;
    for (int64_t base = lo + ((int64_t)blockIdx.x * SL_NNM_WAVES + wave) * 16; base < hi; base += step) {
        int64_t idx = base + (lane & 15);
        const bool valid = idx < hi;
        idx = valid ? idx : hi - 1;
        double x[SL_P];
        sl_index_to_grid_point(M.m.grid, M.gf, d, idx, x);
        double v = nn_mfma_eval<NL>(net, wl, lane, x, d, false, nullptr);
        if (M.m.value.negate) v = v * -1.0;
        if (valid && lane < 16) values[idx - lo] = v;
    }
}

template <int NL, int DT, int MT>
__global__ __launch_bounds__(64 * SL_NNM_WAVES) void k_nn_check_mfma(
    const SlDevModel M, SlAux aux, int64_t lo, int64_t hi, const uint64_t* __restrict__ init_bits,
    const double* __restrict__ values, const double* __restrict__ records,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    const double* __restrict__ points) {
    extern __shared__ __attribute__((aligned(16))) double wl[];
    __shared__ uint64_t red_v[SL_NNM_WAVES];
    __shared__ int64_t red_i[SL_NNM_WAVES];
    const SlNet& net = *aux.net;
    nn_stage_weights(net, wl);
    const SlDims n = sl_dims<DT, MT>(M);
    const int d = n.d, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lvk = M.m.lipschitz.lv_kind;
    const bool grad_lv = lvk == SL_LIP_ABS_GRAD || lvk == SL_LIP_NORM_GRAD;
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    // a wavefront owns one 64-cell word of the mask: four tiles of 16 cells
    const int64_t step = (int64_t)gridDim.x * SL_NNM_WAVES * 64;
    for (int64_t wbase = lo + ((int64_t)blockIdx.x * SL_NNM_WAVES + wave) * 64; wbase < hi; wbase += step) {
        uint64_t word = 0ull;
        const int64_t widx = (wbase - lo) >> 6;
        const uint64_t init = init_bits ? init_bits[widx] : 0ull;
        for (int tile = 0; tile < 4; ++tile) {
            const int64_t base = wbase + 16 * tile;
            if (base >= hi) break;
            int64_t idx = base + (lane & 15);
            const bool valid = idx < hi;
            idx = valid ? idx : hi - 1;
            double x[SL_P], u[SL_M], nxt[SL_D], err[SL_D], lv_x[SL_D], lv_n[SL_D], g[SL_D];
            sl_cell_state(M, d, idx, points, x);
            if (records) {
                const double* r = records + (idx - lo) * (2 + 2 * d);
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d) { nxt[k] = r[2 + k]; err[k] = r[2 + d + k]; }
            } else {
                sl_policy_any<true>(M, n, aux.tri, idx, x, u);
                sl_append_action(n, u, x);
                sl_dynamics_det<0>(M, n, x, nxt);
            }
            double v_x = nn_mfma_eval<NL>(net, wl, lane, x, d, grad_lv, g);
            if (grad_lv) nn_lv_from_grad(lvk, d, g, lv_x); else sl_lv(M, d, x, lv_x);
            const bool grad_n = grad_lv && M.uncertain;
            double v_n = nn_mfma_eval<NL>(net, wl, lane, nxt, d, grad_n, g);
            if (M.uncertain) { if (grad_lv) nn_lv_from_grad(lvk, d, g, lv_n); else sl_lv(M, d, nxt, lv_n); }
            if (M.m.value.negate) { v_x = v_x * -1.0; v_n = v_n * -1.0; }
            const double decrease = sl_decrease(M, d, v_x, v_n, lv_n, err);
            const double threshold = sl_threshold(M, d, lv_x, M.m.lipschitz.tau, x);
            const bool negative = valid && (decrease < threshold);
            if (valid && dbg && lane < 16) {
                double* o = dbg + (idx - lo) * (2 + 2 * d);
                o[0] = decrease; o[1] = threshold;
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < d) { o[2 + k] = nxt[k]; o[2 + d + k] = M.uncertain ? err[k] : 0.0; }
            }
            const uint64_t bits = __ballot(negative) & 0xffffull;
            word |= bits << (16 * tile);
            const bool ok = negative || ((init >> (16 * tile + (lane & 15))) & 1ull);
            if (valid && !ok && lane < 16) {
                const double key_v = values ? values[idx - lo] : v_x;
                sl_key_min(best_v, best_i, sl_vbits(key_v), idx);
            }
        }
        if (lane == 0) neg_bits[widx] = word;
    }
    sl_block_reduce_key<true>(best_v, best_i, red_v, red_i);
    if (threadIdx.x == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

#define SL_NN_DISPATCH(KERNEL, ...)                                                               \
    do {                                                                                          \
        const size_t wl = sizeof(double) * (size_t)ctx->h_net.wtotal;                             \
        const int nl = ctx->h_net.nlayers;                                                        \
        auto k1 = KERNEL(1); auto k2 = KERNEL(2); auto k3 = KERNEL(3); auto k4 = KERNEL(4);       \
        auto kern = nl == 1 ? k1 : (nl == 2 ? k2 : (nl == 3 ? k3 : k4));                          \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)wl));                                          \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * SL_NNM_WAVES), wl, ctx->stream, \
                           __VA_ARGS__);                                                          \
    } while (0)

int sl_nn_values_launch(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_values) {
    int64_t blocks = (hi - lo + 16 * SL_NNM_WAVES - 1) / (16 * SL_NNM_WAVES);
    const int64_t cap = (int64_t)ctx->num_cu * (ctx->h_net.wtotal * 8 > 75 * 1024 ? 1 : 2);
    if (blocks > cap) blocks = cap;
    SlAux aux{ctx->d_tri, ctx->d_net};
#define SL_NN_VALUES(NL_) k_nn_values_mfma<NL_>
    SL_NN_DISPATCH(SL_NN_VALUES, ctx->h_model, aux, lo, hi, d_values);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

int sl_nn_check_launch(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                   const double* d_values, const double* d_records, uint64_t* d_neg_bits,
                   int* nblocks, double* d_dbg, const double* d_points) {
    int64_t blocks = (hi - lo + 64 * SL_NNM_WAVES - 1) / (64 * SL_NNM_WAVES);
    int64_t cap = (int64_t)ctx->num_cu * (ctx->h_net.wtotal * 8 > 75 * 1024 ? 1 : 2);
    if (cap > SL_MAX_GRID) cap = SL_MAX_GRID;
    if (blocks > cap) blocks = cap;
    *nblocks = (int)blocks;
    SlAux aux{ctx->d_tri, ctx->d_net};
    const int variant = sl_dim_variant_of(ctx->h_model);
#define SL_NN_CHECK_D2(NL_) k_nn_check_mfma<NL_, 2, 1>
#define SL_NN_CHECK_D4(NL_) k_nn_check_mfma<NL_, 4, 1>
#define SL_NN_CHECK_D0(NL_) k_nn_check_mfma<NL_, 0, 0>
#define SL_NN_CHECK_ARGS ctx->h_model, aux, lo, hi, d_init_bits, d_values, d_records, d_neg_bits, \
                     ctx->d_partials, d_dbg, d_points
    if (variant == 2) SL_NN_DISPATCH(SL_NN_CHECK_D2, SL_NN_CHECK_ARGS);
    else if (variant == 4) SL_NN_DISPATCH(SL_NN_CHECK_D4, SL_NN_CHECK_ARGS);
    else SL_NN_DISPATCH(SL_NN_CHECK_D0, SL_NN_CHECK_ARGS);
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, d_records != nullptr, "k_nn_check_mfma<layers=%d, d=%d>", ctx->h_net.nlayers,
                   variant);
    return SL_OK;
}
