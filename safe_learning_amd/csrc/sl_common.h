// sl_common.h - context object and device helpers shared by the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "sl_hip.h"
#include "sl_model.h"

#define SL_BLOCK 256
#define SL_MAX_GRID 2048

struct SlGpHeadHost {
    bool set = false;
    int n = 0, n_pad = 0, p = 0, dout = 0, col0 = 0, cfg = 0;
    double lengthscales[SL_MAX_INPUT_DIM] = {};   // as uploaded: appended points are scaled by the
                                                  // same division as the packed ones (bit-identical)
    double* d_xs = nullptr;
    double* d_mpack = nullptr;
    double* d_alpha = nullptr;
    sl_gp_kernel* d_kernel = nullptr;             // sum-of-products kernel of the head, or null (RBF)
    sl_gp_kernel h_kernel;
};

struct sl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool model_set = false;

    SlDevModel h_model;            // host copy
    SlDevModel* d_model = nullptr; // device copy read by the kernels
    SlGpDev h_gp;
    SlGpDev* d_gp = nullptr;
    SlGpHeadHost gp_heads[SL_MAX_GP_HEADS];
    int gp_cfg = 0;

    SlTri h_tri[2];
    SlTri* d_tri = nullptr;        // [2]
    double* d_tri_points[2] = {nullptr, nullptr};
    SlNet h_net;
    SlNet* d_net = nullptr;
    double* d_net_kernels = nullptr;

    void* d_scratch = nullptr;         // grown on demand (sl_eval_points)
    size_t scratch_bytes = 0;
    double* d_gp4_seeds = nullptr;     // k_gp_sweep4: seeds of the k_x sequences, per workgroup
    size_t gp4_seed_bytes = 0;
    void* d_records = nullptr;         // GP posterior records of the two-pass network check
    size_t records_bytes = 0;
    sl_key* d_partials = nullptr;      // SL_MAX_GRID entries x 4 keys
    int64_t* d_partial_counts = nullptr;
    unsigned long long* d_ticket = nullptr;   // tile counter of k_gp_small (zeroed per launch)
    double* d_actions = nullptr;       // bellman action list
    int num_cu = 256;
    void* comm = nullptr;              // RCCL communicator (sl_comm.hip), optional
    void* d_comm_records = nullptr;    // [world] gathered sl_sweep_result records
    int comm_rank = 0, comm_world = 1;
    char last_kernel[160] = "";        // dominant kernel(s) of the last sweep call (sl_last_kernel)
    // k_bellman4_policy: what is derived from the policy alone (its action at every cell, the
    // distinct values, the tile order) is kept between sweeps as long as the policy is the same
    void* d_policy_cache = nullptr;
    size_t policy_cache_bytes = 0;
    unsigned long long policy_token = 0;   // bumped by sl_model_set (policy / grid changed), sl_tri_set
                                           // and sl_tri_set_table of slot 1
    struct {
        bool valid = false;
        int64_t lo = 0, hi = 0;
        unsigned long long token = 0;
        int n_glob = 0;
        unsigned long long bits[64];
    } policy_cache;
};

extern thread_local std::string g_sl_last_error;

int sl_fail(sl_ctx* ctx, int code, const char* fmt, ...);
// name of the kernel a sweep entry point launched (append = true: "a + b" for two-pass sweeps)
void sl_note_kernel(sl_ctx* ctx, bool append, const char* fmt, ...);

// launchers defined in the other translation units
int sl_gp_sweep_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                       const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                       int* nblocks, double* d_dbg, const double* d_points);
int sl_gp4_sweep_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                        const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                        int* nblocks, double* d_dbg, const double* d_points);
bool sl_gp4_supports(const SlDevModel& model);
bool sl_det_rows_supports(const SlDevModel& model, int64_t lo, int64_t hi);
int sl_det_rows_launch(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                       const double* d_values, uint64_t* d_neg_bits, int* nblocks);
bool sl_gp_small_supports(sl_ctx* ctx, const SlDevModel& model);
int sl_gp_small_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                       const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                       int* nblocks, double* d_dbg, const double* d_points);
int sl_bellman4_launch(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, double* d_v_new,
                       int32_t* d_argmax, double* d_q, double* d_stats, int* done);
int sl_bellman4_policy_launch(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_v_new, double* d_stats,
                              int* done);
int sl_nn_values_launch(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_values);
int sl_nn_check_launch(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                       const double* d_values, const double* d_records, uint64_t* d_neg_bits,
                       int* nblocks, double* d_dbg, const double* d_points);

#define SL_HIP_CHECK(ctx, call)                                                             \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return sl_fail((ctx), SL_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                           hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

// The model (and the GP head table) travel BY VALUE in the kernel-argument segment: every field
// is then fetched with scalar loads and used as an SGPR operand.  Only the large tables
// (triangulation, network) stay behind pointers.
struct SlAux {
    const SlTri* __restrict__ tri;  // [2]
    const SlNet* __restrict__ net;
};
static_assert(sizeof(SlDevModel) + sizeof(SlGpDev) < 3900, "kernel arguments must stay below 4 KiB");

// ---- wave / block reductions on (V, index) keys -----------------------------------------
__device__ __forceinline__ void sl_key_min(uint64_t& v, int64_t& i, uint64_t ov, int64_t oi) {
    if (sl_key_less(ov, oi, v, i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void sl_key_max(uint64_t& v, int64_t& i, uint64_t ov, int64_t oi) {
    if (sl_key_less(v, i, ov, oi)) { v = ov; i = oi; }
}

template <bool IS_MIN>
__device__ __forceinline__ void sl_wave_reduce_key(uint64_t& v, int64_t& i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t ov = __shfl_xor((unsigned long long)v, off, 64);
        int64_t oi = __shfl_xor((long long)i, off, 64);
        if (IS_MIN) sl_key_min(v, i, ov, oi); else sl_key_max(v, i, ov, oi);
    }
}

// Reduce over the whole block; result valid in thread 0. `sv`/`si` are LDS scratch of nwaves.
template <bool IS_MIN>
__device__ __forceinline__ void sl_block_reduce_key(uint64_t& v, int64_t& i, uint64_t* sv,
                                                    int64_t* si) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    sl_wave_reduce_key<IS_MIN>(v, i);
    __syncthreads();
    if (lane == 0) { sv[wave] = v; si[wave] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            if (IS_MIN) sl_key_min(v, i, sv[w], si[w]); else sl_key_max(v, i, sv[w], si[w]);
        }
    }
}

// state of cell `idx`: generated from the flat grid index, or read from an explicit point list
__device__ __forceinline__ void sl_cell_state(const SlDevModel& M, int d, int64_t idx,
                                              const double* __restrict__ points, double* x) {
    if (points) {
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) x[k] = points[idx * d + k];
    } else {
        sl_index_to_state(M.m.grid, M.gf, d, idx, x);
    }
}

// ---- the per-cell Lyapunov check shared by the deterministic and the GP kernels --------------
struct SlCellCheck {
    double v_x, decrease, threshold;
    bool negative;
};

// policy(x).  GENERAL = false: closed-form and per-vertex table policies only (no scratch).
template <bool GENERAL>
__device__ __forceinline__ void sl_policy_any(const SlDevModel& M, SlDims n, const SlTri* tri,
                                              int64_t idx, const double* x, double* u) {
    const sl_policy_desc& p = M.m.policy;
    if (p.kind == SL_POLICY_TABLE) {
#pragma unroll
        for (int a = 0; a < SL_M; ++a) if (a < n.m) u[a] = p.d_table[idx * n.m + a];
        sl_saturate(p, n.m, u);
    } else if (GENERAL && p.kind == SL_POLICY_TRI) {
#pragma unroll
        for (int a = 0; a < SL_M; ++a) if (a < n.m) u[a] = sl_tri_eval(tri[1], x, a, nullptr);
        sl_saturate(p, n.m, u);
    } else {
        sl_policy_closed_form(M, n, x, u);
    }
}

// =============================================================================================
// LyapunovNetwork forward / input gradient (examples/utilities.py:85-104)
// =============================================================================================
__device__ inline double sl_network_value(const SlNet& net, const double* z, int d, double* grad) {
    double act[SL_MAX_NN_LAYERS + 1][SL_NN_MAXW];
    double pre[SL_MAX_NN_LAYERS][SL_NN_MAXW];
    for (int k = 0; k < d; ++k) act[0][k] = z[k];
    for (int l = 0; l < net.nlayers; ++l) {
        const int in = net.dims[l], out = net.dims[l + 1];
        const double* K = net.kernels + net.koff[l];
        for (int o = 0; o < out; ++o) {
            double s = 0.0;
            for (int i = 0; i < in; ++i) s = fma(act[l][i], K[o * in + i], s);
            pre[l][o] = s;
            act[l + 1][o] = sl_act(net.act[l], s);
        }
    }
    const int last = net.dims[net.nlayers];
    double value = 0.0;
    for (int o = 0; o < last; ++o) value = fma(act[net.nlayers][o], act[net.nlayers][o], value);
    if (grad) {
        double g[SL_NN_MAXW], gn[SL_NN_MAXW];
        for (int o = 0; o < last; ++o) g[o] = 2.0 * act[net.nlayers][o];
        for (int l = net.nlayers - 1; l >= 0; --l) {
            const int in = net.dims[l], out = net.dims[l + 1];
            const double* K = net.kernels + net.koff[l];
            for (int i = 0; i < in; ++i) gn[i] = 0.0;
            for (int o = 0; o < out; ++o) {
                double t = g[o] * sl_dact(net.act[l], pre[l][o], act[l + 1][o]);
                for (int i = 0; i < in; ++i) gn[i] = fma(t, K[o * in + i], gn[i]);
            }
            for (int i = 0; i < in; ++i) g[i] = gn[i];
        }
        for (int k = 0; k < d; ++k) grad[k] = g[k];
    }
    return value;
}


// Copy a table descriptor (simplices, hyperplanes, grid: ~6 KB) into the workgroup's LDS.  The
// lookups walk every unit-cell simplex; read from global memory each walk is a chain of load
// round trips that the compiler cannot hoist (the kernels store in between).
__device__ __forceinline__ void sl_stage_tri(SlTri* dst, const SlTri* src) {
    static_assert(sizeof(SlTri) % 4 == 0, "SlTri is copied word by word");
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(SlTri) / 4); i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// Both table slots in LDS for the "general" flavours of the sweeps (nothing for the others).
template <bool ON> struct SlTriLds { SlTri t[2]; };
template <> struct SlTriLds<false> { int unused; };
template <bool ON>
__device__ __forceinline__ SlAux sl_stage_aux(SlTriLds<ON>& lds, const SlAux& aux) {
    if constexpr (ON) {
        static_assert(sizeof(SlTri) % 4 == 0, "SlTri is copied word by word");
        const uint32_t* s = reinterpret_cast<const uint32_t*>(aux.tri);
        uint32_t* d = reinterpret_cast<uint32_t*>(lds.t);
        for (int i = threadIdx.x; i < (int)(2 * sizeof(SlTri) / 4); i += blockDim.x) d[i] = s[i];
        __syncthreads();
        return SlAux{lds.t, aux.net};
    } else {
        return aux;
    }
}

// Flavours of the value / L_v code: SL_FAST quadratic V only (no scratch), SL_TABLES adds the
// interpolated table (V = -value_function of the RL loop, |tri.gradient| as L_v), SL_FULL also the
// per-thread LyapunovNetwork (4.6 KB of private arrays per lane).  The grid sweeps never need
// SL_FULL: a network V is routed to the matrix-core kernels of sl_nn.hip, so their "general"
// instantiations are compiled as SL_TABLES and carry no network scratch; only the point
// evaluation of sl_eval_points uses SL_FULL.
enum { SL_FAST = 0, SL_FULL = 1, SL_TABLES = 2 };
template <bool GENERAL> struct SlSweepFlavour { static constexpr int value = GENERAL ? SL_TABLES : SL_FAST; };

// V(z)
template <int G>
__device__ __forceinline__ double sl_value_any(const SlDevModel& M, int d, const SlAux& aux,
                                               const double* z) {
    if (G == SL_FAST || M.m.value.kind == SL_V_QUADRATIC) return sl_quadratic(M.m.value, d, z);
    if (G == SL_TABLES || M.m.value.kind == SL_V_TRI) {
        double v = sl_tri_eval(aux.tri[0], z, 0, nullptr);
        return M.m.value.negate ? (v * -1.0) : v;
    }
    double v = sl_network_value(*aux.net, z, d, nullptr);
    return M.m.value.negate ? (v * -1.0) : v;
}

// L_v(z) including |grad V|
template <int G>
__device__ __forceinline__ void sl_lv_any(const SlDevModel& M, int d, const SlAux& aux,
                                          const double* z, double* lv) {
    const int kind = M.m.lipschitz.lv_kind;
    if (G == SL_FAST || (kind != SL_LIP_ABS_GRAD && kind != SL_LIP_NORM_GRAD)) { sl_lv(M, d, z, lv); return; }
    double g[SL_D];
    if (G == SL_TABLES || M.m.value.kind == SL_V_TRI) sl_tri_eval(aux.tri[0], z, 0, g);
    else sl_network_value(*aux.net, z, d, g);
    if (M.m.value.negate) {
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) g[k] = g[k] * -1.0;
    }
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

// V(z) and L_v(z) together: a network value function shares one forward pass between the value
// and its input gradient (lyapunov_function_learning.ipynb cell 19: L_v = |grad V|_1).
template <int G>
__device__ __forceinline__ double sl_value_and_lv(const SlDevModel& M, int d, const SlAux& aux,
                                                  const double* z, double* lv) {
    const int kind = M.m.lipschitz.lv_kind;
    if (G == SL_FULL && M.m.value.kind == SL_V_NETWORK &&
        (kind == SL_LIP_ABS_GRAD || kind == SL_LIP_NORM_GRAD)) {
        double g[SL_D];
        double v = sl_network_value(*aux.net, z, d, g);
        if (M.m.value.negate) v = v * -1.0;          // |grad| is sign independent
        if (kind == SL_LIP_ABS_GRAD) {
#pragma unroll
            for (int k = 0; k < SL_D; ++k) if (k < d) lv[k] = fabs(g[k]);
        } else {
            double acc = fabs(g[0]);
#pragma unroll
            for (int k = 1; k < SL_D; ++k) if (k < d) acc = acc + fabs(g[k]);
            lv[0] = acc;
        }
        return v;
    }
    sl_lv_any<G>(M, d, aux, z, lv);
    return sl_value_any<G>(M, d, aux, z);
}

template <int G>
__device__ __forceinline__ SlCellCheck sl_cell_check(const SlDevModel& M, int d, const SlAux& aux,
                                                     const double* x, const double* next_mean,
                                                     const double* err) {
    SlCellCheck r;
    double lv_x[SL_D], lv_n[SL_D];
    r.v_x = sl_value_and_lv<G>(M, d, aux, x, lv_x);
    double v_next;
    if (M.uncertain) v_next = sl_value_and_lv<G>(M, d, aux, next_mean, lv_n);
    else v_next = sl_value_any<G>(M, d, aux, next_mean);
    r.decrease = sl_decrease(M, d, r.v_x, v_next, lv_n, err);
    r.threshold = sl_threshold(M, d, lv_x, M.m.lipschitz.tau, x);
    r.negative = r.decrease < r.threshold;
    return r;
}

// blocks of SL_BLOCK threads that walk `ncells` items with a grid stride
static inline int sl_grid_blocks(int64_t ncells) {
    int64_t b = (ncells + SL_BLOCK - 1) / SL_BLOCK;
    if (b > SL_MAX_GRID) b = SL_MAX_GRID;
    if (b < 1) b = 1;
    return (int)b;
}

// kernel-variant id of a model: 0 = generic, 1..4 = (d, 1)
static inline int sl_dim_variant_of(const SlDevModel& M) {
    if (M.m.policy.m == 1 && M.m.grid.d >= 1 && M.m.grid.d <= 4) return M.m.grid.d;
    return 0;
}

// true when the model needs the table / network code paths
static inline bool sl_model_is_general(const SlDevModel& M) {
    return M.m.value.kind != SL_V_QUADRATIC || M.m.policy.kind == SL_POLICY_TRI ||
           M.m.lipschitz.lv_kind == SL_LIP_ABS_GRAD || M.m.lipschitz.lv_kind == SL_LIP_NORM_GRAD;
}
