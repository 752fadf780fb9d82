// sl_succ.hip - successor cache of the Bellman sweeps.
//
// reinforcement_learning.py:89-104: a sweep evaluates  r(x_i, u_a) + gamma V(f(x_i, u_a))  where only
// the LAST step reads the value table - the next state f(x_i, u_a) (for GP dynamics: the posterior
// mean, an FP64 GEMM over all training points) and the place where it falls in the value grid do
// not depend on V.  A value-iteration loop repeats the sweep hundreds of times with the same
// dynamics and action set (470 sweeps to a 1e-6 residual at 64^4 x 9 actions), so the first max sweep
// keeps, per (vertex, action), what the interpolation of functions.py:1473-1499 needs:
//
//     corner   int32   vertex index of the rectangle's lower corner      (functions.py:754-776)
//     simplex  uint8   unit-cell simplex                                 (functions.py:1103-1158)
//     w[1..D]  f64     barycentric weights of the simplex's vertices 1..D (functions.py:1180-1200)
//
// = 8 D + 5 bytes per pair (37 at D = 4; 6.2 GB for 64^4 x (9 actions + the vertex itself), of a
// GPU with 288 GB), struct-of-arrays so that a wavefront's loads are contiguous.  The later sweeps
// (k_bellman_cached) rebuild the located point bit for bit (sl_tri_reloc's rule: w[0] by the same
// ordered sum), gather the D + 1 table values and combine them with the same fused multiply-adds as
// sl_tri_value_fast: the tables they produce are IDENTICAL to the uncached sweeps'
// (tests/test_gpu_rl.py::test_successor_cache_*), at the cost of reading the cache - HBM-bound,
// 8 D + 5 + gathers instead of 2 n D flops per pair.
//
// Policy evaluation (n_actions == 0) with a table policy whose value at every vertex is one of the
// cached actions (the greedy policies of the loop) selects that action's entry: k_succ_select maps
// every vertex to its action once per policy (keys rounded to 2^-40 like k_bellman4_policy), the
// vertex's own interpolated value V(x_i) of the Bellman error comes from slot A.
//
// Validity: ctx->dynamics_token (grid, dynamics description, GP heads, structure of the value
// triangulation - bumped by sl_model_set / sl_gp_* / sl_tri_set(0)), the range and the action set.
// The value table, the reward, gamma and the policy may change freely.  A cache that does not fit
// its budget is simply not built (the sweeps then recompute, as before).
#include "sl_common.h"

namespace {

constexpr size_t HEADER = 1024;          // the action list, at the head of the allocation

size_t succ_bytes(int slots, int d, int64_t n) {
    const size_t w = sizeof(double) * (size_t)slots * d * n;
    const size_t c = (sizeof(int32_t) * (size_t)slots * n + 15) & ~(size_t)15;
    const size_t s = ((size_t)slots * n + 15) & ~(size_t)15;
    return HEADER + w + c + s;
}

SlSuccDev view_of(void* base, int slots, int d, int64_t n) {
    SlSuccDev v;
    char* b = reinterpret_cast<char*>(base) + HEADER;
    v.w = reinterpret_cast<double*>(b);
    b += sizeof(double) * (size_t)slots * d * n;
    v.corner = reinterpret_cast<int32_t*>(b);
    b += (sizeof(int32_t) * (size_t)slots * n + 15) & ~(size_t)15;
    v.simplex = reinterpret_cast<uint8_t*>(b);
    v.actions = reinterpret_cast<const double*>(base);
    v.n = n;
    v.slots = slots;
    v.d = d;
    return v;
}

int64_t budget_of(const sl_ctx* ctx) {
    if (ctx->succ.max_bytes >= 0) return ctx->succ.max_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    return (int64_t)(total_b / 4);
}

}  // namespace

SlSuccDev sl_succ_view(const sl_ctx* ctx) {
    const auto& S = ctx->succ;
    if (!S.d) {
        SlSuccDev v;
        memset(&v, 0, sizeof(v));
        return v;
    }
    return view_of(S.d, S.n_actions + 1, S.d_state, S.hi - S.lo);
}

SlSuccDev sl_succ_begin_fill(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, const double* h_actions) {
    SlSuccDev none;
    memset(&none, 0, sizeof(none));
    auto& S = ctx->succ;
    S.valid = false;
    S.select_valid = false;
    const SlDevModel& M = ctx->h_model;
    const int variant = sl_dim_variant_of(M);                 // d = 1 .. 4 with one action dimension
    if (!S.enabled || variant == 0 || n_actions < 1 || n_actions > SL_MAX_ACTIONS || hi <= lo) return none;
    if (M.gf.nindex > 0x7fffffffll) return none;              // corners are int32
    const int64_t n = hi - lo;
    const size_t need = succ_bytes(n_actions + 1, variant, n);
    if ((int64_t)need > budget_of(ctx)) return none;
    if (need > S.bytes) {
        if (S.d) (void)hipFree(S.d);
        S.d = nullptr;
        S.bytes = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need + ((size_t)1 << 30) > free_b) return none;
        if (hipMalloc(&S.d, need) != hipSuccess) {
            (void)hipGetLastError();                          // out of memory is not an error here
            S.d = nullptr;
            return none;
        }
        S.bytes = need;
    }
    S.lo = lo;
    S.hi = hi;
    S.n_actions = n_actions;
    S.d_state = variant;
    S.m = 1;
    S.token = ctx->dynamics_token;
    memset(S.actions, 0, sizeof(S.actions));
    memcpy(S.actions, h_actions, sizeof(double) * (size_t)n_actions);
    if (hipMemcpyAsync(S.d, S.actions, sizeof(double) * SL_MAX_ACTIONS, hipMemcpyHostToDevice, ctx->stream) !=
        hipSuccess) {
        (void)hipGetLastError();
        return none;
    }
    return sl_succ_view(ctx);
}

void sl_succ_commit(sl_ctx* ctx) {
    ctx->succ.valid = true;
    ++ctx->succ.fills;
}

// ---------------------------------------------------------------------------------------------
// the sweeps from the cache
// ---------------------------------------------------------------------------------------------
// One cache entry: what the interpolation of a located point needs (sl_tri_reloc's rule with the
// vertex offsets of the simplex from LDS: the same rows, the same weights, bit for bit).  The two
// halves are separate so that a kernel can issue the loads of several entries, then their
// gathers, before it combines the first (memory-level parallelism: the sweep is bound by the
// latency of these two dependent round trips unless several are in flight).
template <int D>
struct SuccEntry {
    int64_t corner;
    int simplex;
    double w[D];
};
template <int D>
__device__ __forceinline__ void succ_load(const SlSuccDev& sc, int slot, int64_t cell, SuccEntry<D>& e) {
    // (read once per sweep: non-temporal, so that the cache lines of the value table - gathered
    // below, 134 MB at 64^4 - are the ones L2 and the Infinity Cache keep)
    e.corner = __builtin_nontemporal_load(&sc.corner[(int64_t)slot * sc.n + cell]);
    e.simplex = __builtin_nontemporal_load(&sc.simplex[(int64_t)slot * sc.n + cell]);
#pragma unroll
    for (int j = 0; j < D; ++j) e.w[j] = __builtin_nontemporal_load(&sc.w[((int64_t)slot * D + j) * sc.n + cell]);
}
template <int D>
__device__ __forceinline__ void succ_gather(const SuccEntry<D>& e, const int64_t* voff, int ncols,
                                            const double* __restrict__ table, double* vals) {
#pragma unroll
    for (int j = 0; j <= D; ++j) vals[j] = table[(e.corner + voff[e.simplex * (D + 1) + j]) * ncols];
}
template <int D>
__device__ __forceinline__ double succ_combine(const SuccEntry<D>& e, const double* vals) {
    SlTriLoc<D> loc;
    double wsum = 0.0;
#pragma unroll
    for (int j = 1; j <= D; ++j) {
        loc.w[j] = e.w[j - 1];
        wsum += e.w[j - 1];
    }
    loc.w[0] = 1.0 - wsum;
    return sl_tri_combine<D>(loc, vals);
}
// (Measured and dropped, round 6, 64^4 x 9: one entry at a time 1.35 - 1.42 ms; three entries in
// flight - shipped - 1.40 ms; two cells per thread with 16-byte loads 1.40 ms: the kernel moves its
// 6.3 GB - rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, profiles/r06_C5_cached_traffic.txt, against 5.9 GB
// of entries + 0.2 GB of outputs - at 4.5 TB/s next to the gathers, whatever its instruction mix.)
// POLICY = false: max over the cached actions (value_iteration(action_space),
//                 discrete_policy_optimization; reinforcement_learning.py:266-279)
// POLICY = true:  the cached action the policy takes at the vertex (value_iteration(), :135-140;
//                 bellmann_error, :116-133)
template <int D, bool POLICY>
__global__ __launch_bounds__(256) void k_bellman_cached(
    const SlDevModel M, SlAux aux, const SlSuccDev sc, int64_t lo, int64_t hi, int n_actions,
    const double* __restrict__ usel, const int8_t* __restrict__ asel, double* __restrict__ v_new,
    int32_t* __restrict__ argmax, double* __restrict__ q_out, double* __restrict__ stats) {
    __shared__ int64_t voff[SL_MAX_SIMPLICES * (D + 1)];
    __shared__ double act_l[SL_MAX_ACTIONS];
    __shared__ double red_max[4], red_sum[4];
    const SlTri& vt = aux.tri[0];                              // scalar loads of the few fields used
    for (int t = threadIdx.x; t < vt.nsimplex * (D + 1); t += blockDim.x) {
        const int code = vt.simplices[t / (D + 1)][t % (D + 1)];
        int64_t v = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) v += ((code >> k) & 1) * vt.stride[k];
        voff[t] = v;
    }
    if ((int)threadIdx.x < SL_MAX_ACTIONS)
        act_l[threadIdx.x] = sc.actions[threadIdx.x];
    __syncthreads();
    const SlDims nd = sl_dims<D, 1>(M);
    const int p = nd.p, A = n_actions, ncols = vt.ncols;
    const double* __restrict__ table = vt.table;
    double lmax = 0.0, lsum = 0.0;
    for (int64_t idx = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = idx - lo;
        double x[SL_P], u[SL_M];
        sl_index_to_state(M.m.grid, M.gf, D, idx, x);
        double best_q = 0.0, v_self = 0.0;
        int best_a = -1;
        if (POLICY) {
            if (asel[cell] < 0) continue;                      // k_succ_policy_miss computes it
            u[0] = usel[cell];
            sl_append_action(nd, u, x);
            const double r = sl_quadratic(M.m.reward, p, x);
            // the vertex's action and the vertex itself (slot A): both entries in flight together
            SuccEntry<D> ea, es;
            double va[D + 1], vs[D + 1];
            succ_load<D>(sc, asel[cell], cell, ea);
            succ_load<D>(sc, A, cell, es);
            succ_gather<D>(ea, voff, ncols, table, va);
            succ_gather<D>(es, voff, ncols, table, vs);
            double v = succ_combine<D>(ea, va);
            v_self = succ_combine<D>(es, vs);
            if (M.m.value.negate) v = v * -1.0;
            const double tq = M.m.gamma * v;
            best_q = r + tq;
        } else {
            // three actions at a time: their entries, then their 3 (D + 1) gathers, are in flight
            // together; the maximum is still taken in ascending action order (first maximiser wins)
            constexpr int G = 3;
            for (int a0 = 0; a0 < A; a0 += G) {
                SuccEntry<D> e[G];
                double vals[G][D + 1];
#pragma unroll
                for (int g = 0; g < G; ++g) succ_load<D>(sc, a0 + g < A ? a0 + g : A - 1, cell, e[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) succ_gather<D>(e[g], voff, ncols, table, vals[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int a = a0 + g;
                    if (a < A) {
                        u[0] = act_l[a];
                        sl_append_action(nd, u, x);
                        const double r = sl_quadratic(M.m.reward, p, x);
                        double v = succ_combine<D>(e[g], vals[g]);
                        if (M.m.value.negate) v = v * -1.0;
                        const double tq = M.m.gamma * v;
                        const double q = r + tq;               // reinforcement_learning.py:104
                        if (q_out) q_out[cell * A + a] = q;
                        if (best_a < 0 || q > best_q) { best_q = q; best_a = a; }
                    }
                }
            }
            if (argmax) argmax[cell] = best_a;
        }
        v_new[cell] = best_q;
        double v_old = table[idx * ncols];
        if (M.m.value.negate) v_old = v_old * -1.0;
        lmax = fmax(lmax, fabs(best_q - v_old));
        if (POLICY) {
            double v_int = v_self;
            if (M.m.value.negate) v_int = v_int * -1.0;
            const double diff = best_q - v_int;
            lsum = fma(diff, diff, lsum);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
        lsum += __shfl_xor(lsum, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red_max[threadIdx.x >> 6] = lmax; red_sum[threadIdx.x >> 6] = lsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lmax = fmax(lmax, red_max[w]); lsum += red_sum[w]; }
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        if (POLICY) atomicAdd(&stats[1], lsum);
    }
}

// The policy's value at every vertex of [lo, hi) and the cached action it is (sl_b4_action_bits:
// an interpolated table read at its own vertices returns the vertex value up to the rounding of
// the barycentric weights).  Vertices whose value is none of the cached actions - where the
// reference's interpolation rule extrapolates from a neighbouring simplex (functions.py:1103-1158:
// rectangle by digitize, simplex by x mod unit_maxes) - go to the miss list (at most `cap` of them
// are recorded, misses[0] counts all): k_succ_policy_miss evaluates them one by one.
#ifndef SL_SUCC_SELECT_WAVES
#define SL_SUCC_SELECT_WAVES 4
#endif
template <int DT>
__global__ __launch_bounds__(256, SL_SUCC_SELECT_WAVES) void k_succ_select(const SlDevModel M, SlAux aux, const SlSuccDev sc,
                                                     int64_t lo, int64_t hi, int n_actions,
                                                     double* __restrict__ usel, int8_t* __restrict__ asel,
                                                     unsigned long long* __restrict__ misses,
                                                     int32_t* __restrict__ miss_list, int64_t cap) {
    __shared__ SlTriLds<true> tri_l;
    aux = sl_stage_aux<true>(tri_l, aux);
    __shared__ unsigned long long abits[SL_MAX_ACTIONS];
    if ((int)threadIdx.x < SL_MAX_ACTIONS) abits[threadIdx.x] = sl_b4_action_bits(sc.actions[threadIdx.x]);
    __syncthreads();
    const SlDims nd = sl_dims<DT, 1>(M);
    for (int64_t idx = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * blockDim.x) {
        double x[SL_P], u[SL_M];
        sl_index_to_state(M.m.grid, M.gf, nd.d, idx, x);
        sl_policy_any<true>(M, nd, aux.tri, idx, x, u);
        const unsigned long long b = sl_b4_action_bits(u[0]);
        int a = -1;
        for (int s = n_actions - 1; s >= 0; --s) a = abits[s] == b ? s : a;
        usel[idx - lo] = u[0];
        asel[idx - lo] = (int8_t)a;
        if (a < 0) {
            const unsigned long long slot = atomicAdd(misses, 1ull);
            if ((int64_t)slot < cap) miss_list[slot] = (int32_t)(idx - lo);
        }
    }
}

// Policy evaluation at the vertices of the miss list: the successor of (x_i, u_i) computed directly
// and located / interpolated as usual.  A WAVEFRONT per vertex: its lanes share the training points
// of the posterior mean (reinforcement_learning.py:98-99; any kernel family), partial sums folded
// across the lanes - 825 such vertices at 64^4 x 9 actions, one thread each walking 1024 points was
// 0.7 ms of a 1.3 ms sweep.
template <int DT>
__global__ __launch_bounds__(256) void k_succ_policy_miss(
    const SlDevModel M, const SlGpDev gp, SlAux aux, const SlSuccDev sc, int64_t lo, int64_t nmiss,
    const int32_t* __restrict__ miss_list, const double* __restrict__ usel,
    double* __restrict__ v_new, double* __restrict__ stats) {
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p, lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double lmax = 0.0, lsum = 0.0;
    for (int64_t t = wave; t < nmiss; t += nwaves) {
        const int64_t cell = miss_list[t], idx = lo + cell;
        double x[SL_P], u[SL_M], nxt[SL_D];
        sl_index_to_state(M.m.grid, M.gf, d, idx, x);
        u[0] = usel[cell];
        sl_append_action(nd, u, x);
        if (M.m.dynamics.kind == SL_DYN_GP) {
#pragma unroll
            for (int k = 0; k < SL_D; ++k) nxt[k] = 0.0;
            for (int h = 0; h < gp.nheads; ++h) {
                const SlGpHeadDev& hd = gp.head[h];
                double xg[SL_P];
#pragma unroll
                for (int qd = 0; qd < SL_P; ++qd) xg[qd] = (qd < p) ? x[qd] * hd.inv_ls[qd] : 0.0;
                for (int j = lane; j < hd.n; j += 64) {
                    double z = 0.0, xa[SL_P];
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
                        if (k < d && dd >= 0 && dd < hd.dout) nxt[k] = fma(kx, hd.alpha[j * hd.dout + dd], nxt[k]);
                    }
                }
            }
            double prior[SL_D];
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d) {
                    for (int o = 32; o >= 1; o >>= 1) nxt[k] += __shfl_xor(nxt[k], o, 64);
                    nxt[k] = nxt[k] + prior[k];
                }
            }
        } else {
            sl_dynamics_det<0>(M, nd, x, nxt);
        }
        const double r = sl_quadratic(M.m.reward, p, x);
        double v = sl_tri_value_fast<DT>(vt, nxt);
        if (M.m.value.negate) v = v * -1.0;
        const double tq = M.m.gamma * v;
        const double q = r + tq;
        double v_old = vt.table[idx * vt.ncols];
        double v_int = sl_tri_value_fast<DT>(vt, x);
        if (M.m.value.negate) { v_old = v_old * -1.0; v_int = v_int * -1.0; }
        if (lane == 0) {
            v_new[cell] = q;
            lmax = fmax(lmax, fabs(q - v_old));
            const double diff = q - v_int;
            lsum = fma(diff, diff, lsum);
        }
    }
    if (lane == 0 && (lmax > 0.0 || lsum > 0.0)) {
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        atomicAdd(&stats[1], lsum);
    }
}

int sl_succ_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, const double* h_actions,
                  double* d_v_new, int32_t* d_argmax, double* d_q, double* d_stats, int* done) {
    *done = 0;
    auto& S = ctx->succ;
    if (!S.enabled || !S.valid || !S.d || S.token != ctx->dynamics_token || S.lo != lo || S.hi != hi)
        return SL_OK;
    const SlDevModel& M = ctx->h_model;
    const int variant = sl_dim_variant_of(M);
    if (variant != S.d_state || M.m.policy.m != S.m) return SL_OK;
    const SlSuccDev sc = sl_succ_view(ctx);
    const int64_t n = hi - lo;
    SlAux aux{ctx->d_tri, ctx->d_net};
    const int64_t nblk = (n + 255) / 256;
    const int blocks = (int)(nblk < 32 * (int64_t)ctx->num_cu ? nblk : 32 * (int64_t)ctx->num_cu);
    const bool policy = n_actions == 0;
    double* usel = nullptr;
    int8_t* asel = nullptr;
    int32_t* miss_list = nullptr;
    int64_t nmiss = 0;
    if (!policy) {
        if (n_actions != S.n_actions ||
            memcmp(h_actions, S.actions, sizeof(double) * (size_t)n_actions) != 0)
            return SL_OK;
    } else {
        // closed-form policies take values outside any finite set; a per-vertex table or an
        // interpolated table (the greedy policies of the loop) is matched against the cached actions
        const int pk = M.m.policy.kind;
        if (pk != SL_POLICY_TRI && pk != SL_POLICY_TABLE) return SL_OK;
        // [n] policy values, [n] action indices, the miss counter, the miss list (n / 8 + 64 entries:
        // a policy that misses more often than that is not the loop's greedy table)
        const int64_t cap = n / 8 + 64;
        const size_t off_asel = sizeof(double) * (size_t)n;
        const size_t off_count = off_asel + (((size_t)n + 15) & ~(size_t)15);
        const size_t off_list = off_count + 16;
        const size_t need = off_list + sizeof(int32_t) * (size_t)cap;
        if (need > S.select_bytes) {
            if (S.d_select) (void)hipFree(S.d_select);
            S.d_select = nullptr;
            S.select_bytes = 0;
            S.select_valid = false;
            if (hipMalloc(&S.d_select, need) != hipSuccess) {
                (void)hipGetLastError();
                S.d_select = nullptr;
                return SL_OK;
            }
            S.select_bytes = need;
        }
        char* base = reinterpret_cast<char*>(S.d_select);
        usel = reinterpret_cast<double*>(base);
        asel = reinterpret_cast<int8_t*>(base + off_asel);
        unsigned long long* misses = reinterpret_cast<unsigned long long*>(base + off_count);
        miss_list = reinterpret_cast<int32_t*>(base + off_list);
        if (!(S.select_valid && S.select_policy_token == ctx->policy_token && S.select_token == S.token &&
              S.select_lo == lo && S.select_hi == hi)) {
            S.select_valid = false;
            SL_HIP_CHECK(ctx, hipMemsetAsync(misses, 0, 8, ctx->stream));
#define SL_SELECT(D_)                                                                              \
    hipLaunchKernelGGL(k_succ_select<D_>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->h_model,  \
                       aux, sc, lo, hi, S.n_actions, usel, asel, misses, miss_list, cap)
            if (variant == 4) SL_SELECT(4); else if (variant == 3) SL_SELECT(3);
            else if (variant == 2) SL_SELECT(2); else SL_SELECT(1);
#undef SL_SELECT
            SL_HIP_CHECK(ctx, hipGetLastError());
            unsigned long long h_miss = 0;
            SL_HIP_CHECK(ctx, hipMemcpyAsync(&h_miss, misses, 8, hipMemcpyDeviceToHost, ctx->stream));
            SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            S.select_valid = true;
            S.select_misses = (int64_t)h_miss;
            S.select_usable = (int64_t)h_miss <= cap;
            S.select_policy_token = ctx->policy_token;
            S.select_token = S.token;
            S.select_lo = lo;
            S.select_hi = hi;
        }
        if (!S.select_usable) return SL_OK;
        nmiss = S.select_misses;
    }
#define SL_CACHED(D_, P_)                                                                          \
    hipLaunchKernelGGL((k_bellman_cached<D_, P_>), dim3(blocks), dim3(256), 0, ctx->stream,        \
                       ctx->h_model, aux, sc, lo, hi, S.n_actions, usel, asel, d_v_new, d_argmax,  \
                       d_q, d_stats)
#define SL_CACHED_D(P_)                                                                            \
    do {                                                                                           \
        if (variant == 4) SL_CACHED(4, P_); else if (variant == 3) SL_CACHED(3, P_);               \
        else if (variant == 2) SL_CACHED(2, P_); else SL_CACHED(1, P_);                            \
    } while (0)
    if (policy) SL_CACHED_D(true); else SL_CACHED_D(false);
#undef SL_CACHED_D
#undef SL_CACHED
    SL_HIP_CHECK(ctx, hipGetLastError());
    if (policy && nmiss > 0) {
        const int64_t mblk = (nmiss + 3) / 4;                  // a wavefront per vertex
        const int mblocks = (int)(mblk < 8 * (int64_t)ctx->num_cu ? mblk : 8 * (int64_t)ctx->num_cu);
#define SL_MISS(D_)                                                                                \
    hipLaunchKernelGGL(k_succ_policy_miss<D_>, dim3(mblocks), dim3(256), 0, ctx->stream,           \
                       ctx->h_model, ctx->h_gp, aux, sc, lo, nmiss, miss_list, usel, d_v_new, d_stats)
        if (variant == 4) SL_MISS(4); else if (variant == 3) SL_MISS(3);
        else if (variant == 2) SL_MISS(2); else SL_MISS(1);
#undef SL_MISS
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    if (policy) ++S.policy_hits; else ++S.hits;
    if (policy && nmiss > 0)
        sl_note_kernel(ctx, false, "k_bellman_cached<d=%d, policy> (successor cache, %d actions, %lld vertices "
                       "outside the action set one by one)", variant, S.n_actions, (long long)nmiss);
    else
        sl_note_kernel(ctx, false, "k_bellman_cached<d=%d, %s> (successor cache, %d actions)", variant,
                       policy ? "policy" : "max", S.n_actions);
    *done = 1;
    return SL_OK;
}

extern "C" int sl_successor_cache_configure(sl_ctx* ctx, int64_t max_bytes) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_successor_cache_configure: NULL context");
    auto& S = ctx->succ;
    S.max_bytes = max_bytes < 0 ? -1 : max_bytes;
    S.enabled = max_bytes != 0;
    if (max_bytes >= 0 && (int64_t)S.bytes > max_bytes) {
        SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (S.d) (void)hipFree(S.d);
        if (S.d_select) (void)hipFree(S.d_select);
        S.d = S.d_select = nullptr;
        S.bytes = S.select_bytes = 0;
        S.valid = S.select_valid = false;
    }
    return SL_OK;
}

extern "C" int sl_successor_cache_info(sl_ctx* ctx, sl_successor_cache_stats* out) {
    if (!ctx || !out) return sl_fail(ctx, SL_ERR_INVALID, "sl_successor_cache_info: NULL argument");
    const auto& S = ctx->succ;
    memset(out, 0, sizeof(*out));
    out->bytes = (int64_t)S.bytes;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    out->max_bytes = S.enabled ? budget_of(ctx) : 0;
    out->valid = (S.enabled && S.valid && S.token == ctx->dynamics_token) ? 1 : 0;
    out->lo = S.lo;
    out->hi = S.hi;
    out->n_actions = S.n_actions;
    out->fills = S.fills;
    out->hits = S.hits;
    out->policy_hits = S.policy_hits;
    return SL_OK;
}
