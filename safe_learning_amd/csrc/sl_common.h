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
#define SL_TIMING_CHANNELS 3
#define SL_MAX_ACTIONS 16                 // actions of a Bellman max sweep (sl_bellman_sweep)
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

// Environment switches (A/B runs and the tests of the alternative kernels; none is needed in
// production).  Read ONCE, by sl_ctx_create - a launch never calls getenv.  -1 = not set.
struct SlEnv {
    int gp_cfg = -1;                 // SL_GP_CFG=0..3: force a GP-sweep configuration
    int gp_small = -1;               // SL_GP_SMALL=0: small training sets stay on k_gp_sweep
    int gp_small_waves = -1;         // SL_GP_SMALL_WAVES=8
    int gp_small_split = -1;         // SL_GP_SMALL_SPLIT
    int det_rows = -1;               // SL_DET_ROWS=0: linear dynamics on k_det_sweep
    int gp4_one_panel = -1;          // SL_GP4_ONE_PANEL=0: 193..256 points stay on k_gp_small
    int gp4_seeds = -1;              // SL_GP4_SEEDS=0: every k_x chunk from the exponentials
    int gp4_tickets = -1;            // SL_GP4_TICKETS=0/1: fixed tile list / tile counter
    int bellman_mfma = -1;           // SL_BELLMAN_MFMA=0: Bellman sweeps on the FP64-VALU kernel
    int bellman4 = -1, bellman4_policy = -1, bellman4_policy_cache = -1, bellman4_policy_verbose = -1;
    int bellman4_ragged = -1, bellman4_quarter = -1, bellman4_split = -1, bellman4_round = -1;
    int bellman4_shared = -1;        // SL_BELLMAN4*: see DESIGN.md "Environment switches"
    int probe_blocks_per_cu = -1;    // SL_PROBE_BLOCKS_PER_CU (sl_debug_fp64_rate)
    int succ_cache = -1;             // SL_SUCC_CACHE=0: no successor cache
    // (The switches that make kernels SKIP work - SL_GP4_SKIP, SL_BM_FLAGS, SL_B4P_FLAGS: timing
    // attribution, results meaningless - exist in development builds of a unit only, -DSL_DIAG by
    // tools/build_variant.sh: sl_diag_flags below.  The shipped library cannot run a sweep that
    // leaves work out.)
};
#ifdef SL_DIAG
static inline int sl_diag_flags(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}
#else
static inline constexpr int sl_diag_flags(const char*) { return 0; }
#endif
void sl_env_read(SlEnv* env);

// A NeuralNetwork policy (sl_policy_net.hip): per layer the dense kernel [in][out] and an optional
// bias at params + koff / boff (boff < 0: none), activation codes 0 linear, 1 tanh, 2 relu, 3 sigmoid.
struct SlPolicyNet {
    int32_t nlayers, set;
    int32_t dims[SL_MAX_NN_LAYERS + 1];
    int32_t act[SL_MAX_NN_LAYERS];
    int32_t koff[SL_MAX_NN_LAYERS], boff[SL_MAX_NN_LAYERS];
    double scale;
    const double* params;
};

struct sl_ctx {
    int device = 0;
    SlEnv env;
    SlPolicyNet pnet = {};
    double* d_pnet_params = nullptr;
    void* d_policy_actions = nullptr;      // per-cell action table of a network policy (one call's cells)
    size_t policy_actions_bytes = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool model_set = false;

    SlDevModel h_model;            // host copy
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
    unsigned long long* d_ticket = nullptr;   // [0]: tile counter of k_gp_small (zeroed per launch); the
                                              // first word of [1]: finished workgroups of k_finalize_dev
                                              // (its last workgroup reduces and resets it)
    double* d_actions = nullptr;       // bellman action list
    int num_cu = 256;
    void* comm = nullptr;              // RCCL communicator (sl_comm.hip), optional
    void* d_comm_records = nullptr;    // [world] gathered sl_sweep_result records
    int comm_rank = 0, comm_world = 1;
    char last_kernel[160] = "";        // dominant kernel(s) of the last sweep call (sl_last_kernel)
    // sl_timing_configure: pairs of HIP events around the launches of an entry point, on the stream
    // they go to (channel 0: sl_lyap_sweep, 1: sl_lyap_finalize_dev, 2: sl_bellman_sweep)
    struct Timing {
        int slots = 0;
        int used[SL_TIMING_CHANNELS] = {0, 0, 0};
        hipEvent_t* events[SL_TIMING_CHANNELS] = {nullptr, nullptr, nullptr};   // [2 * slots] each
    } timing;
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
    // Successor cache of the Bellman sweeps (sl_succ.hip): where the next state of every (vertex,
    // action) pair of [lo, hi) lies in the value grid - rectangle corner, unit-cell simplex and
    // barycentric weights.  None of it depends on the value table
    // (reinforcement_learning.py:89-104), so the sweeps after the first of a value-iteration loop
    // only gather and combine.  Valid while dynamics_token (grid, dynamics description, GP heads,
    // structure of the value triangulation) and the action set are what they were at fill time.
    unsigned long long dynamics_token = 0;
    struct SuccState {
        bool enabled = true;               // SL_SUCC_CACHE=0 at sl_ctx_create, sl_successor_cache_configure
        int64_t max_bytes = 0;             // budget (default: a quarter of the device's memory)
        void* d = nullptr;                 // [A + 1][D][n] weights, [A + 1][n] corners, [A + 1][n] simplices
        size_t bytes = 0;
        bool valid = false;
        int64_t lo = 0, hi = 0;
        int n_actions = 0, d_state = 0, m = 0;
        unsigned long long token = 0;
        double actions[SL_MAX_ACTIONS * SL_MAX_ACTION_DIM];
        // policy evaluation from the cache: the cached action of every vertex, under policy_token
        void* d_select = nullptr;          // [n] policy values (double), [n] action index (int8)
        size_t select_bytes = 0;
        bool select_valid = false, select_usable = false;
        int64_t select_misses = 0, select_lo = 0, select_hi = 0;
        unsigned long long select_policy_token = 0, select_token = 0;
        long long fills = 0, hits = 0, policy_hits = 0;     // sl_successor_cache_info
        bool filling = false, filled = false;               // inside one sl_bellman_sweep
    } succ;
};

// device view of the successor cache of n = hi - lo vertices (slot a < A: action a, slot A: the
// vertex itself - the interpolated V(x_i) of the Bellman error)
struct SlSuccDev {
    double* w;                 // [(slot * D + j) * n + cell], j = 0 .. D-1: weights 1 .. D
    int32_t* corner;           // [slot * n + cell]
    uint8_t* simplex;          // [slot * n + cell]
    const double* actions;     // [SL_MAX_ACTIONS] the action set the cache was filled for
    int64_t n;
    int32_t slots, d;
};
SlSuccDev sl_succ_view(const sl_ctx* ctx);
// the cache to fill during this max sweep over [lo, hi) (w == nullptr: none - disabled, over the
// budget, shapes the cache does not take); marks the cache invalid until sl_succ_commit
SlSuccDev sl_succ_begin_fill(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, const double* h_actions);
void sl_succ_commit(sl_ctx* ctx);
// *done = 1 when the sweep was served from the cache
int sl_succ_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, const double* h_actions,
                  double* d_v_new, int32_t* d_argmax, double* d_q, double* d_stats, int* done);

// Brackets an entry point's launches with a pair of events when sl_timing_configure asked for it
// (bench.py's kernel durations: recorded by the library on its own stream, no host objects per call)
struct SlTimed {
    sl_ctx* ctx;
    int channel, slot;
    SlTimed(sl_ctx* c, int ch) : ctx(c), channel(ch), slot(-1) {
        if (!c || c->timing.slots <= c->timing.used[ch]) return;
        slot = c->timing.used[ch];
        (void)hipEventRecord(c->timing.events[ch][2 * slot], c->stream);
    }
    ~SlTimed() {
        if (slot < 0) return;
        (void)hipEventRecord(ctx->timing.events[channel][2 * slot + 1], ctx->stream);
        ctx->timing.used[channel] = slot + 1;
    }
};

// Brackets an entry point that evaluates the policy: a network policy (SL_POLICY_NETWORK) is
// evaluated once per cell of [lo, hi) (or per explicit point) into an action table and the model
// the kernels see carries SL_POLICY_TABLE on it until the scope ends.  rc != SL_OK: give up.
struct SlPolicyTableScope {
    sl_ctx* ctx;
    bool swapped;
    int rc;
    sl_policy_desc saved;
    SlPolicyTableScope(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_points);
    ~SlPolicyTableScope();
    SlPolicyTableScope(const SlPolicyTableScope&) = delete;
    SlPolicyTableScope& operator=(const SlPolicyTableScope&) = delete;
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

// A policy value rounded to 2^-40: the key under which the policy-evaluation kernels group the
// values an interpolated table policy takes at its own vertices (sl_bellman4.hip, sl_succ.hip).
__device__ __forceinline__ unsigned long long sl_b4_action_bits(double u) {
    const double c = fabs(u) < 256.0 ? __builtin_rint(u * 0x1p40) * 0x1p-40 : u;
    return (unsigned long long)__double_as_longlong(c + 0.0);
}

// sl_tri_value_fast that also leaves the located point in the successor cache (slot, cell) when
// there is one to fill: the value is computed from the SAME located point either way.
template <int DT>
__device__ __forceinline__ double sl_tri_value_fill(const SlTri& vt, const double* z, const SlSuccDev& sc,
                                                    int slot, int64_t cell) {
    if constexpr (DT == 0) {
        return sl_tri_eval(vt, z, 0, nullptr);
    } else {
        SlTriLoc<DT> loc;
        double vals[DT + 1];
        sl_tri_locate_fast<DT>(vt, z, loc);
        sl_tri_gather<DT>(vt, loc, vals);
        if (sc.w) {
#pragma unroll
            for (int j = 0; j < DT; ++j) sc.w[((int64_t)slot * DT + j) * sc.n + cell] = loc.w[j + 1];
            sc.corner[(int64_t)slot * sc.n + cell] = (int32_t)loc.corner;
            sc.simplex[(int64_t)slot * sc.n + cell] = (uint8_t)loc.simplex;
        }
        return sl_tri_combine<DT>(loc, vals);
    }
}
// only the located point (the vertex itself: slot A)
template <int DT>
__device__ __forceinline__ void sl_tri_fill_only(const SlTri& vt, const double* z, const SlSuccDev& sc,
                                                 int slot, int64_t cell) {
    if constexpr (DT > 0) {
        if (sc.w) {
            SlTriLoc<DT> loc;
            sl_tri_locate_fast<DT>(vt, z, loc);
#pragma unroll
            for (int j = 0; j < DT; ++j) sc.w[((int64_t)slot * DT + j) * sc.n + cell] = loc.w[j + 1];
            sc.corner[(int64_t)slot * sc.n + cell] = (int32_t)loc.corner;
            sc.simplex[(int64_t)slot * sc.n + cell] = (uint8_t)loc.simplex;
        }
    }
}

// Next state of one (x, u) pair, z = [x, u]: the GP posterior MEAN (reinforcement_learning.py:98-99;
// one training point at a time, any kernel family) plus the prior mean, or the deterministic
// dynamics.  The one-pair-at-a-time path of k_bellman and of the successor cache's policy misses.
__device__ __forceinline__ void sl_next_state_mean(const SlDevModel& M, const SlGpDev& gp, SlDims nd,
                                                   const double* x, double* nxt) {
    const int d = nd.d, p = nd.p;
    if (M.m.dynamics.kind == SL_DYN_GP) {
        double prior[SL_D];
        sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) nxt[k] = 0.0;
        for (int h = 0; h < gp.nheads; ++h) {
            const SlGpHeadDev& hd = gp.head[h];
            double xg[SL_P];
#pragma unroll
            for (int qd = 0; qd < SL_P; ++qd) xg[qd] = (qd < p) ? x[qd] * hd.inv_ls[qd] : 0.0;
#pragma unroll 4
            for (int j = 0; j < hd.n; ++j) {
                double z = 0.0;
                double xa[SL_P];
#pragma unroll
                for (int qd = 0; qd < SL_P; ++qd) {
                    xa[qd] = 0.0;
                    if (qd < p) {
                        xa[qd] = hd.xs[qd * hd.n_pad + j];
                        const double dlt = xa[qd] - xg[qd];
                        z = fma(dlt, dlt, z);
                    }
                }
                const double kx = hd.kernel ? sl_kernel_eval(*hd.kernel, p, xa, xg)
                                            : hd.variance * sl_exp_nonpos(-0.5 * z);
#pragma unroll
                for (int k = 0; k < SL_D; ++k) {
                    const int dd = k - hd.col0;
                    if (k < d && dd >= 0 && dd < hd.dout)
                        nxt[k] = fma(kx, hd.alpha[j * hd.dout + dd], nxt[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) nxt[k] = nxt[k] + prior[k];
    } else {
        sl_dynamics_det<0>(M, nd, x, nxt);
    }
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
