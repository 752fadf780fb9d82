// sl_nn.hip - Lyapunov check and value pass for a LyapunovNetwork value function
// (examples/utilities.py:85-104; config C3 of BASELINE.json).
//
// A workgroup of SL_NN_WAVES wavefronts owns 64 grid cells (lane = cell); the wavefronts split the
// output features of every layer among themselves.  The activations of every layer live in LDS
// as [feature][cell] (conflict-free 512-byte rows), the layer kernels are read with wave-uniform
// addresses (scalar loads, SGPR operands of v_fma_f64), eight outputs are register blocked per
// LDS read, one barrier per layer.  The backward pass for the input gradient (L_v = |grad V|, the notebooks'
// tf.gradients) reuses the activation buffers in place, so one forward + backward costs
// 2 * sum(in_l * out_l) FMAs per cell and no scratch memory.
//
// The posterior (mean, error) of GP dynamics comes from a first pass of k_gp_sweep that only
// emits its per-cell records; deterministic dynamics are evaluated here.
#include "sl_common.h"

#define SL_NN_WAVES 8
#define SL_NN_BLOCK (64 * SL_NN_WAVES)
#define SL_NN_OB 8                       // outputs per register block

// value (and optionally d value / d input) of the network at one point per lane.
// act: LDS [total features][64]; z: this lane's input; grad may be null.
__device__ __forceinline__ double nn_eval(const SlNet& net, double* __restrict__ act, int lane,
                                          int wave, const double* z, int d, double* grad) {
    const double* __restrict__ kernels = net.kernels;
    // layer l reads P[l] (offset poff[l]) and writes P[l+1]; P[0] = input
    int poff[SL_MAX_NN_LAYERS + 1];
    poff[0] = 0;
    for (int l = 0; l < net.nlayers; ++l) poff[l + 1] = poff[l] + net.dims[l];
    __syncthreads();                                   // previous use of the buffers is over
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) act[k * 64 + lane] = z[k];
    }
    __syncthreads();
    for (int l = 0; l < net.nlayers; ++l) {
        const int in = net.dims[l], out = net.dims[l + 1], a = net.act[l];
        const double* __restrict__ K = net.kernels_t + net.koff[l];    // [in][out]
        const double* src = act + poff[l] * 64;
        double* dst = act + poff[l + 1] * 64;
        for (int o = wave * SL_NN_OB; o < out; o += SL_NN_OB * SL_NN_WAVES) {
            double acc[SL_NN_OB];
#pragma unroll
            for (int k = 0; k < SL_NN_OB; ++k) acc[k] = 0.0;
#pragma unroll 4
            for (int i = 0; i < in; ++i) {
                const double h = src[i * 64 + lane];
#pragma unroll
                for (int k = 0; k < SL_NN_OB; ++k)
                    if (o + k < out) acc[k] = fma(h, K[i * out + o + k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < SL_NN_OB; ++k)
                if (o + k < out) dst[(o + k) * 64 + lane] = sl_act(a, acc[k]);
        }
        __syncthreads();
    }
    const int L = net.nlayers, last = net.dims[L];
    double value = 0.0;
    {
        const double* top = act + poff[L] * 64;
        for (int o = 0; o < last; ++o) {
            const double h = top[o * 64 + lane];
            value = fma(h, h, value);
        }
    }
    if (!grad) return value;
    // backward: t_L = 2 h_L act'(h_L) stored in place of P[L]; then layer by layer
    __syncthreads();                                   // every wavefront has summed P[L]
    {
        double* top = act + poff[L] * 64;
        const int a = net.act[L - 1];
        for (int o = wave; o < last; o += SL_NN_WAVES) {
            const double h = top[o * 64 + lane];
            top[o * 64 + lane] = 2.0 * h * sl_dact(a, h > 0.0 ? 1.0 : -1.0, h);
        }
    }
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
        const int in = net.dims[l], out = net.dims[l + 1];
        const double* __restrict__ K = kernels + net.koff[l];
        const double* t = act + poff[l + 1] * 64;
        double* dst = act + poff[l] * 64;
        const int a_prev = l > 0 ? net.act[l - 1] : 0;
        for (int i = wave * SL_NN_OB; i < in; i += SL_NN_OB * SL_NN_WAVES) {
            double acc[SL_NN_OB];
#pragma unroll
            for (int k = 0; k < SL_NN_OB; ++k) acc[k] = 0.0;
#pragma unroll 4
            for (int o = 0; o < out; ++o) {
                const double tv = t[o * 64 + lane];
#pragma unroll
                for (int k = 0; k < SL_NN_OB; ++k)
                    if (i + k < in) acc[k] = fma(tv, K[o * in + i + k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < SL_NN_OB; ++k) {
                if (i + k < in) {
                    if (l > 0) {
                        const double h = dst[(i + k) * 64 + lane];
                        dst[(i + k) * 64 + lane] = acc[k] * sl_dact(a_prev, h > 0.0 ? 1.0 : -1.0, h);
                    } else {
                        dst[(i + k) * 64 + lane] = acc[k];
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < SL_D; ++k) if (k < d) grad[k] = act[k * 64 + lane];
    return value;
}

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

// values[i] = V(all_points[i])
__global__ __launch_bounds__(SL_NN_BLOCK) void k_nn_values(const SlDevModel M, SlAux aux, int64_t lo,
                                                           int64_t hi, double* __restrict__ values) {
    extern __shared__ __attribute__((aligned(16))) double act[];
    const SlNet& net = *aux.net;
    const int d = M.m.grid.d, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int64_t base = lo + (int64_t)blockIdx.x * 64; base < hi; base += (int64_t)gridDim.x * 64) {
        int64_t idx = base + lane;
        const bool valid = idx < hi;
        idx = valid ? idx : hi - 1;
        double x[SL_P];
        sl_index_to_grid_point(M.m.grid, M.gf, d, idx, x);
        double v = nn_eval(net, act, lane, wave, x, d, nullptr);
        if (M.m.value.negate) v = v * -1.0;
        if (valid && wave == 0) values[idx - lo] = v;
    }
}

// decrease check with a network V.  records: optional [cells][2+2d] with the GP posterior
// (mean at [2..2+d), error at [2+d..2+2d)) from the first pass.
__global__ __launch_bounds__(SL_NN_BLOCK) void k_nn_check(
    const SlDevModel M, SlAux aux, int64_t lo, int64_t hi, const uint64_t* __restrict__ init_bits,
    const double* __restrict__ values, const double* __restrict__ records,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    const double* __restrict__ points) {
    extern __shared__ __attribute__((aligned(16))) double act[];
    const SlNet& net = *aux.net;
    const SlDims n = sl_dims<0, 0>(M);
    const int d = n.d, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lvk = M.m.lipschitz.lv_kind;
    const bool grad_lv = lvk == SL_LIP_ABS_GRAD || lvk == SL_LIP_NORM_GRAD;
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    for (int64_t base = lo + (int64_t)blockIdx.x * 64; base < hi; base += (int64_t)gridDim.x * 64) {
        int64_t idx = base + lane;
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
        double v_x = nn_eval(net, act, lane, wave, x, d, grad_lv ? g : nullptr);
        if (grad_lv) nn_lv_from_grad(lvk, d, g, lv_x); else sl_lv(M, d, x, lv_x);
        double v_n = nn_eval(net, act, lane, wave, nxt, d, (grad_lv && M.uncertain) ? g : nullptr);
        if (M.uncertain) { if (grad_lv) nn_lv_from_grad(lvk, d, g, lv_n); else sl_lv(M, d, nxt, lv_n); }
        if (M.m.value.negate) { v_x = v_x * -1.0; v_n = v_n * -1.0; }
        const double decrease = sl_decrease(M, d, v_x, v_n, lv_n, err);
        const double threshold = sl_threshold(M, d, lv_x, M.m.lipschitz.tau);
        const bool negative = valid && (decrease < threshold);
        if (wave != 0) continue;                       // every wavefront holds the same results
        if (valid && dbg) {
            double* o = dbg + (idx - lo) * (2 + 2 * d);
            o[0] = decrease; o[1] = threshold;
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d) { o[2 + k] = nxt[k]; o[2 + d + k] = M.uncertain ? err[k] : 0.0; }
        }
        const uint64_t word = __ballot(negative);
        const int64_t widx = (base - lo) >> 6;
        if (lane == 0) neg_bits[widx] = word;
        const uint64_t init = init_bits ? init_bits[widx] : 0ull;
        const bool ok = negative || ((init >> lane) & 1ull);
        if (valid && !ok) {
            const double key_v = values ? values[idx - lo] : v_x;
            sl_key_min(best_v, best_i, sl_vbits(key_v), idx);
        }
    }
    sl_wave_reduce_key<true>(best_v, best_i);
    if (threadIdx.x == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

static size_t nn_lds_bytes(const SlNet& n) {
    size_t feats = 0;
    for (int l = 0; l <= n.nlayers; ++l) feats += n.dims[l];
    return feats * 64 * sizeof(double);
}

int sl_nn_values_launch(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_values) {
    const size_t lds = nn_lds_bytes(ctx->h_net);
    if (lds > 160 * 1024)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "network too wide for LDS staging (%zu bytes)", lds);
    SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_values),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (hi - lo + 63) / 64;
    const int64_t cap = (int64_t)ctx->num_cu * (lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3));
    if (blocks > cap) blocks = cap;
    SlAux aux{ctx->d_tri, ctx->d_net};
    hipLaunchKernelGGL(k_nn_values, dim3((unsigned)blocks), dim3(SL_NN_BLOCK), lds, ctx->stream,
                       ctx->h_model, aux, lo, hi, d_values);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

int sl_nn_check_launch(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                       const double* d_values, const double* d_records, uint64_t* d_neg_bits,
                       int* nblocks, double* d_dbg, const double* d_points) {
    const size_t lds = nn_lds_bytes(ctx->h_net);
    if (lds > 160 * 1024)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "network too wide for LDS staging (%zu bytes)", lds);
    SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_check),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (hi - lo + 63) / 64;
    int64_t cap = (int64_t)ctx->num_cu * (lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3));
    if (cap > SL_MAX_GRID) cap = SL_MAX_GRID;
    if (blocks > cap) blocks = cap;
    *nblocks = (int)blocks;
    SlAux aux{ctx->d_tri, ctx->d_net};
    hipLaunchKernelGGL(k_nn_check, dim3((unsigned)blocks), dim3(SL_NN_BLOCK), lds, ctx->stream,
                       ctx->h_model, aux, lo, hi, d_init_bits, d_values, d_records, d_neg_bits,
                       ctx->d_partials, d_dbg, d_points);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}
