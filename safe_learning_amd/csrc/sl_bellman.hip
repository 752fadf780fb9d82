// sl_bellman.hip - dynamic-programming sweep (reinforcement_learning.py:65-140, 213-279) and
// evaluation of the model's functions at explicit points.
//
// One thread per grid vertex, consecutive lanes = consecutive flat indices (coalesced V / q /
// arg-max stores).  For GP dynamics only the posterior MEAN is needed
// (reinforcement_learning.py:98-99): mean = k_x . alpha' + m(x*), 2 n D flops per (vertex, action).
// With a finite action set the RBF factorises, k_x[j] = S_j(x) * E_j(u_a): the state factor S_j
// (one exp per training point) is shared by all actions, the action factors E_j(u_a) are the same
// for every vertex and are tabulated once per workgroup in LDS.  Training inputs are read with
// wave-uniform addresses, i.e. scalar loads.
#include "sl_common.h"


template <int AMAX, int DT, int MT>
__global__ __launch_bounds__(SL_BLOCK) void k_bellman(
    const SlDevModel M, const SlGpDev gp, SlAux aux, int64_t lo, int64_t hi, int n_actions,
    const double* __restrict__ actions, double* __restrict__ v_new, int32_t* __restrict__ argmax,
    double* __restrict__ q_out, double* __restrict__ stats, const SlSuccDev sc_in) {
    // sc_in.w != 0 (max sweeps only): the located successors go to the successor cache (sl_succ.hip)
    SlSuccDev sc = sc_in;
    if (AMAX == 0) sc.w = nullptr;
    extern __shared__ __attribute__((aligned(16))) double smem[];   // E table [head][n_pad][A]
    __shared__ double red_max[SL_BLOCK / 64], red_sum[SL_BLOCK / 64];
    const SlDims nd = sl_dims<DT, MT>(M);
    const int d = nd.d, m = nd.m, p = nd.p;
    constexpr bool ACTIONS = AMAX > 0;
    const int A = ACTIONS ? n_actions : 1;
    const bool is_gp = M.m.dynamics.kind == SL_DYN_GP;
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;

    // heads with a sum-of-products kernel (sl_gp_set_head_kernel) do not factorise over state and
    // action: every (x, u) is evaluated on its own (the generic path below)
    bool other_kernels = false;
    for (int h = 0; h < gp.nheads; ++h) other_kernels = other_kernels || gp.head[h].kernel != nullptr;
    other_kernels = other_kernels && is_gp;

    // ---- per-workgroup table of action factors E_j(u_a) -------------------------------------
    int e_off[SL_MAX_GP_HEADS];
    if (ACTIONS && is_gp && !other_kernels) {
        int off = 0;
        for (int h = 0; h < gp.nheads; ++h) {
            const SlGpHeadDev& hd = gp.head[h];
            e_off[h] = off;
            // [n_pad][AMAX]; columns a >= A are zero so that the inner loops need no predicate
            for (int t = threadIdx.x; t < hd.n_pad * AMAX; t += SL_BLOCK) {
                const int j = t / AMAX, a = t - j * AMAX;
                double e = 0.0;
                if (a < A) {
                    double z = 0.0;
                    for (int c = 0; c < m; ++c) {
                        const double dlt = hd.xs[(d + c) * hd.n_pad + j] -
                                           actions[a * m + c] * hd.inv_ls[d + c];
                        z = fma(dlt, dlt, z);
                    }
                    e = sl_exp_nonpos(-0.5 * z);
                }
                smem[off + t] = e;
            }
            off += hd.n_pad * AMAX;
        }
        __syncthreads();
    }

    double lmax = 0.0, lsum = 0.0;
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P];
        sl_index_to_state(M.m.grid, M.gf, d, idx, x);

        double best_q = 0.0;
        int best_a = 0;
        if (!ACTIONS || !is_gp || other_kernels) {
            // ---- generic path: one (x, u) at a time --------------------------------------------
            for (int a = 0; a < A; ++a) {
                double u[SL_M], nxt[SL_D];
                if (ACTIONS) {
#pragma unroll
                    for (int c = 0; c < SL_M; ++c) if (c < m) u[c] = actions[a * m + c];
                } else {
                    sl_policy_any<true>(M, nd, aux.tri, idx, x, u);
                }
                sl_append_action(nd, u, x);
                sl_next_state_mean(M, gp, nd, x, nxt);
                const double r = sl_quadratic(M.m.reward, p, x);
                double v = sl_tri_value_fill<DT>(vt, nxt, sc, a, idx - lo);
                if (M.m.value.negate) v = v * -1.0;
                const double t = M.m.gamma * v;
                const double q = r + t;                          // reinforcement_learning.py:104
                if (q_out) q_out[(idx - lo) * A + a] = q;
                if (a == 0 || q > best_q) { best_q = q; best_a = a; }
            }
        } else {
            // ---- GP + action set: state factors shared by all actions ---------------------------
            constexpr int AM = AMAX > 0 ? AMAX : 1;
            double mean[AM][DT > 0 ? DT : SL_D];
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int k = 0; k < (DT > 0 ? DT : SL_D); ++k) mean[a][k] = 0.0;
            for (int h = 0; h < gp.nheads; ++h) {
                const SlGpHeadDev& hd = gp.head[h];
                double xg[SL_D];
#pragma unroll
                for (int k = 0; k < SL_D; ++k) xg[k] = (k < d) ? x[k] * hd.inv_ls[k] : 0.0;
                const double* etab = smem + e_off[h];
#pragma unroll 4
                for (int j = 0; j < hd.n; ++j) {
                    double z = 0.0;
#pragma unroll
                    for (int k = 0; k < SL_D; ++k) {
                        if (k < d) {
                            const double dlt = hd.xs[k * hd.n_pad + j] - xg[k];
                            z = fma(dlt, dlt, z);
                        }
                    }
                    const double s = hd.variance * sl_exp_nonpos(-0.5 * z);
                    // s * alpha'[j][.] once per training point, then one FMA per (action, column)
                    double sa[DT > 0 ? DT : SL_D];
#pragma unroll
                    for (int k = 0; k < (DT > 0 ? DT : SL_D); ++k) {
                        const int dd = k - hd.col0;
                        sa[k] = (dd >= 0 && dd < hd.dout) ? s * hd.alpha[j * hd.dout + dd] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < AM; ++a) {
                        const double e = etab[j * AM + a];
#pragma unroll
                        for (int k = 0; k < (DT > 0 ? DT : SL_D); ++k)
                            mean[a][k] = fma(e, sa[k], mean[a][k]);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < AM; ++a) {
                if (a < A) {
                    double u[SL_M], prior[SL_D], nxt[SL_D];
#pragma unroll
                    for (int c = 0; c < SL_M; ++c) if (c < m) u[c] = actions[a * m + c];
                    sl_append_action(nd, u, x);
                    sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
                    for (int k = 0; k < (DT > 0 ? DT : SL_D); ++k) if (k < d) nxt[k] = mean[a][k] + prior[k];
                    const double r = sl_quadratic(M.m.reward, p, x);
                    double v = sl_tri_value_fill<DT>(vt, nxt, sc, a, idx - lo);
                    if (M.m.value.negate) v = v * -1.0;
                    const double t = M.m.gamma * v;
                    const double q = r + t;
                    if (q_out) q_out[(idx - lo) * A + a] = q;
                    if (a == 0 || q > best_q) { best_q = q; best_a = a; }
                }
            }
        }
        if (ACTIONS) sl_tri_fill_only<DT>(vt, x, sc, A, idx - lo);
        v_new[idx - lo] = best_q;
        if (argmax) argmax[idx - lo] = best_a;
        // stats[0]: max |V_new - V_old| on the vertex table (convergence test of the examples);
        // stats[1]: sum (target - V(x))^2 with V(x) interpolated like the reference does
        //           (reinforcement_learning.py:130-133)
        //           - the Bellman error of the current policy, n_actions == 0 only
        double v_old = vt.table[idx * vt.ncols];
        if (M.m.value.negate) v_old = v_old * -1.0;
        lmax = fmax(lmax, fabs(best_q - v_old));
        if (!ACTIONS) {
            double v_int = sl_tri_value_fast<DT>(vt, x);
            if (M.m.value.negate) v_int = v_int * -1.0;
            const double diff = best_q - v_int;
            lsum = fma(diff, diff, lsum);
        }
    }
    // ---- residual statistics ----------------------------------------------------------------------
    for (int off = 32; off >= 1; off >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, off, 64));
        lsum += __shfl_xor(lsum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red_max[threadIdx.x >> 6] = lmax; red_sum[threadIdx.x >> 6] = lsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SL_BLOCK / 64; ++w) { lmax = fmax(lmax, red_max[w]); lsum += red_sum[w]; }
        // non-negative doubles order like their bit patterns
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        atomicAdd(&stats[1], lsum);
    }
}

// =============================================================================================
// Bellman max sweep on the matrix cores (one shared-input GP head, finite action set)
// =============================================================================================
// mean[cell, (a, dd)] = sum_j S[cell, j] * B[j, (a, dd)]:  the dense K_nm @ alpha contraction of
// reinforcement_learning.py:98-99 as an FP64 MFMA GEMM [cells x n] . [n x A*D] with
//   B[j, (a, dd)] = variance * E_j(u_a) * alpha'[j, dd]       (action factor, packed per sweep)
//   S[cell, j]    = prod_k T_k[i_k(cell)][j]                   (state factor)
// The state factor of the RBF separates over the axes of the product grid, so it is a product of
// d table entries T_k[i][j] = exp(-(x_k(i) - X_jk)^2 / (2 l_k^2)) (N_k x n entries per axis, built
// per sweep by k_bellman_pack) - no exponential in the sweep itself.  A wavefront owns 4 tiles of
// 16 consecutive cells that share every B fragment; lane (c = l & 15, k = l >> 4) produces its own
// A-operand element S[c][4s + k].  The GEMM result goes through LDS to the (cell, action)
// epilogue: prior mean, reward, value-table lookup, arg-max.
typedef double sl_bd4 __attribute__((ext_vector_type(4)));
#define SL_BM_WAVES 8
#define SL_BM_T 4                         // cell tiles per wavefront
// cells per epilogue step of a wavefront (bounded by the LDS the staged means need)
#define SL_BM_SUB_OF(COLBLOCKS) ((COLBLOCKS) > 3 ? 16 : ((COLBLOCKS) == 3 ? 32 : 64))
#define SL_BM_HEADS 4                     // GP heads (FunctionStack members) on the matrix-core path

struct SlBellmanPack {
    int32_t ncb, rowlen, nheads, npad_max; // column blocks per head, padded columns of all heads
    int64_t boff[SL_BM_HEADS];            // start of head h's packed action factors
    int64_t tab0[SL_BM_HEADS];            // start of head h's tables in the pack buffer
    int64_t toff[SL_BM_HEADS][SL_D];      // offset (doubles) of axis k's table inside them
};

__global__ __launch_bounds__(256) void k_bellman_pack(const SlDevModel M, const SlGpDev gp,
                                                      SlBellmanPack pk, int n_actions,
                                                      const double* __restrict__ actions,
                                                      double* __restrict__ pack) {
    const int d = M.m.grid.d, m = M.m.policy.m;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int h = 0; h < pk.nheads; ++h) {
        const SlGpHeadDev& hd = gp.head[h];
        const int nslab = hd.n_pad / 4, ncb = pk.ncb, n_pad = hd.n_pad;
        const int64_t total = (int64_t)nslab * ncb * 64;
        double* bdst = pack + pk.boff[h];
        for (int64_t t = tid; t < total; t += nthreads) {
            const int lane = (int)(t & 63);
            const int cb = (int)((t >> 6) % ncb);
            const int s = (int)((t >> 6) / ncb);
            const int j = 4 * s + (lane >> 4), col = 16 * cb + (lane & 15);
            const int a = col / hd.dout, dd = col - a * hd.dout;
            double v = 0.0;
            if (a < n_actions && j < hd.n) {
                double z = 0.0;
                for (int c = 0; c < m; ++c) {
                    const double dlt = hd.xs[(d + c) * n_pad + j] - actions[a * m + c] * hd.inv_ls[d + c];
                    z = fma(dlt, dlt, z);
                }
                v = hd.variance * sl_exp_nonpos(-0.5 * z) * hd.alpha[j * hd.dout + dd];
            }
            bdst[t] = v;
        }
        int64_t stride = 1;                // flat-index stride of axis k (last axis fastest)
        for (int k = d - 1; k >= 0; --k) {
            const int nk = M.m.grid.num_points[k];
            double* tab = pack + pk.tab0[h] + pk.toff[h][k];
            for (int64_t t = tid; t < (int64_t)nk * n_pad; t += nthreads) {
                // axis d-1: [j][i] (16 consecutive cells read one line); other axes: [i][j]
                const int i = (k == d - 1) ? (int)(t % nk) : (int)(t / n_pad);
                const int j = (k == d - 1) ? (int)(t / nk) : (int)(t % n_pad);
                double x[SL_P];
                sl_index_to_state(M.m.grid, M.gf, d, (int64_t)i * stride, x);
                double v = 0.0;
                if (j < hd.n) {
                    const double dlt = hd.xs[k * n_pad + j] - x[k] * hd.inv_ls[k];
                    v = sl_exp_nonpos(-0.5 * (dlt * dlt));
                }
                tab[t] = v;
            }
            stride *= nk;
        }
    }
}

// NH GP heads (one shared-input head with D outputs, or the members of a FunctionStack with one
// output each), NCB column blocks of (action, output) pairs per head.
template <int DT, int NCB, int NH>
__global__ __launch_bounds__(64 * SL_BM_WAVES) void k_bellman_mfma(
    const SlDevModel M, const SlGpDev gp, SlAux aux, SlBellmanPack pk, int64_t lo, int64_t hi,
    int n_actions, const double* __restrict__ actions, const double* __restrict__ pack,
    double* __restrict__ v_new, int32_t* __restrict__ argmax, double* __restrict__ q_out,
    double* __restrict__ stats, int flags, const SlSuccDev sc) {
#ifndef SL_DIAG
    flags = 0;                                 // the shipped kernel has no diagnostic switches: the tests fold
#endif
    // flags (SL_BM_FLAGS, diagnostics): 1 no GEMM, 2 no (cell, action) epilogue, 4 workgroup
    // barriers instead of wavefront-local ordering
    // sc.w != 0: the located successors go to the successor cache (sl_succ.hip)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_max[SL_BM_WAVES], red_sum[SL_BM_WAVES];
    constexpr int SL_BM_SUB = SL_BM_SUB_OF(NCB * NH);
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p, A = n_actions;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    const int rowlen = pk.rowlen;
    double* my_mean = smem + (size_t)wave * SL_BM_SUB * rowlen;                 // [32 cells][rowlen]
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const int n_last = M.m.grid.num_points[d - 1];
    double lmax = 0.0, lsum = 0.0;
    const int64_t wg_cells = 16 * SL_BM_T * SL_BM_WAVES;
    const int64_t ntiles = (hi - lo + wg_cells - 1) / wg_cells;
    // The staged means and q values are private to a wavefront, whose LDS accesses complete in
    // order: the stages need no workgroup barrier, and the two wavefronts of a SIMD can run one
    // the GEMM and the other the epilogue (integer decode, table gathers) of different tiles.
#define SL_BM_SYNC()                                                                             \
    do {                                                                                         \
        if (flags & 4) __syncthreads();                                                          \
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } \
    } while (0)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = lo + tile * wg_cells + 16 * SL_BM_T * wave;   // first cell of the wavefront
        sl_bd4 acc[NH][SL_BM_T][NCB];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
        const SlGpHeadDev& hd = gp.head[h];
        const int n_pad = hd.n_pad, nslab = n_pad / 4;
        const double* tabs = pack + pk.tab0[h];
        // byte offsets into the tables of this lane's cell in each of the SL_BM_T tiles (element
        // 4s + lk of slab s): 32-bit so that the loads use the scalar-base + vector-offset form
        uint32_t off[SL_BM_T][SL_D];
#pragma unroll
        for (int t = 0; t < SL_BM_T; ++t) {
            int64_t gidx = wbase + 16 * t + lc;
            gidx = gidx < hi ? gidx : hi - 1;
            int64_t ijk[SL_D];
            sl_unravel(M.m.grid, M.gf, d, gidx, ijk);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d - 1) off[t][k] = 8u * (uint32_t)(pk.toff[h][k] + ijk[k] * n_pad + lk);
                else if (k == d - 1) off[t][k] = 8u * (uint32_t)(pk.toff[h][k] + (int64_t)lk * n_last + ijk[k]);
            }
        }
        const uint32_t step_last = 32u * (uint32_t)n_last;       // bytes per slab, axis d-1 ([j][i])
        const char* tabs_b = reinterpret_cast<const char*>(tabs);
#pragma unroll
        for (int t = 0; t < SL_BM_T; ++t)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[h][t][cb] = (sl_bd4){0.0, 0.0, 0.0, 0.0};
        const double* bp = pack + pk.boff[h] + lane;
        double s_cur[SL_BM_T], b_cur[NCB];
#pragma unroll
        for (int t = 0; t < SL_BM_T; ++t) {
            double v = 1.0;
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d) v *= *reinterpret_cast<const double*>(tabs_b + off[t][k]);
            s_cur[t] = v;
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) b_cur[cb] = bp[(size_t)cb * 64];
        // Two slabs per iteration with two named operand sets: the operands of the next slab are
        // requested before, and combined after, the MFMAs of the current one (nslab is even; the
        // look-ahead of the last slab wraps to slab 0 and is discarded).
        double b_odd[NCB], s_odd[SL_BM_T];
#define SL_BM_REQUEST(RAW, BDST, SLAB)                                                          \
        do {                                                                                    \
            const int sl_ = (SLAB) < nslab ? (SLAB) : 0;                                        \
            const double* bq_ = bp + (size_t)sl_ * (NCB * 64);                                  \
            _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) BDST[cb] = bq_[cb * 64];         \
            _Pragma("unroll") for (int t = 0; t < SL_BM_T; ++t) {                               \
                _Pragma("unroll") for (int k = 0; k < SL_D; ++k) {                              \
                    if (k < d)                                                                  \
                        RAW[t][k] = *reinterpret_cast<const double*>(                           \
                            tabs_b + (off[t][k] + (uint32_t)sl_ * (k == d - 1 ? step_last : 32u))); \
                }                                                                               \
            }                                                                                   \
        } while (0)
#define SL_BM_COMBINE(RAW, SDST)                                                                \
        do {                                                                                    \
            _Pragma("unroll") for (int t = 0; t < SL_BM_T; ++t) {                               \
                double v_ = RAW[t][0];                                                          \
                _Pragma("unroll") for (int k = 1; k < SL_D; ++k) if (k < d) v_ *= RAW[t][k];    \
                SDST[t] = v_;                                                                   \
            }                                                                                   \
        } while (0)
#define SL_BM_MFMAS(SSRC, BSRC)                                                                 \
        do {                                                                                    \
            _Pragma("unroll") for (int t = 0; t < SL_BM_T; ++t)                                 \
                _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb)                              \
                    acc[h][t][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(SSRC[t], BSRC[cb],     \
                                                                         acc[h][t][cb], 0, 0, 0); \
        } while (0)
        for (int s = (flags & 1) ? nslab : 0; s < nslab; s += 2) {
            double raw_a[SL_BM_T][SL_D], raw_b[SL_BM_T][SL_D];
            SL_BM_REQUEST(raw_a, b_odd, s + 1);
            SL_BM_MFMAS(s_cur, b_cur);
            SL_BM_COMBINE(raw_a, s_odd);
            SL_BM_REQUEST(raw_b, b_cur, s + 2);
            SL_BM_MFMAS(s_odd, b_odd);
            SL_BM_COMBINE(raw_b, s_cur);
        }
#undef SL_BM_REQUEST
#undef SL_BM_COMBINE
#undef SL_BM_MFMAS
        }
#pragma unroll
        for (int sub = 0; sub < 16 * SL_BM_T / SL_BM_SUB; ++sub) {
            const int64_t sbase = wbase + SL_BM_SUB * sub;
            // D tile: column = lane & 15, row (cell) = (lane >> 4) + 4 * reg
#pragma unroll
            for (int tt = 0; tt < SL_BM_SUB / 16; ++tt) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const sl_bd4 v = acc[h][sub * (SL_BM_SUB / 16) + tt][cb];
                        double* dst = my_mean + (16 * tt + lk) * rowlen + 16 * (h * NCB + cb) + lc;
                        dst[0] = v.x;
                        dst[4 * rowlen] = v.y;
                        dst[8 * rowlen] = v.z;
                        dst[12 * rowlen] = v.w;
                    }
                }
            }
            SL_BM_SYNC();
            // ---- (cell, action) pairs: prior mean, reward, value lookup --------------------------
            // lane = (cell of the step, group of actions): the cell's index decode and state are
            // computed once, each group walks its share of the actions in ascending order
            {
                constexpr int NG = 64 / SL_BM_SUB;
                const int cell = lane & (SL_BM_SUB - 1), grp = lane / SL_BM_SUB;
                const int apg = (A + NG - 1) / NG;
                int64_t idx = sbase + cell;
                const bool live = idx < hi;
                idx = live ? idx : hi - 1;
                double x[SL_P], u[SL_M], prior[SL_D], nxt[SL_D];
                sl_index_to_state(M.m.grid, M.gf, d, idx, x);
                double best_q = 0.0;
                int best_a = -1;
                SlSuccDev sc_l = sc;
                if (!live) sc_l.w = nullptr;
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
                                double mu = 0.0;
#pragma unroll
                                for (int h = 0; h < NH; ++h) {
                                    const int dd = k - gp.head[h].col0, dout = gp.head[h].dout;
                                    if (dd >= 0 && dd < dout)
                                        mu = my_mean[cell * rowlen + 16 * NCB * h + a * dout + dd];
                                }
                                nxt[k] = mu + prior[k];
                            }
                        }
                        const double r = sl_quadratic(M.m.reward, p, x);
                        double v = sl_tri_value_fill<DT>(vt, nxt, sc_l, a, idx - lo);
                        if (M.m.value.negate) v = v * -1.0;
                        const double tq = M.m.gamma * v;
                        const double q = r + tq;
                        if (q_out && live) q_out[(idx - lo) * A + a] = q;
                        if (best_a < 0 || q > best_q) { best_q = q; best_a = a; }
                    }
                }
                // first maximum over the groups in ascending action order
#pragma unroll
                for (int g = 1; g < NG; ++g) {
                    const double oq = __shfl(best_q, cell + g * SL_BM_SUB, 64);
                    const int oa = __shfl(best_a, cell + g * SL_BM_SUB, 64);
                    if (oa >= 0 && (best_a < 0 || oq > best_q)) { best_q = oq; best_a = oa; }
                }
                if (grp == 0 && live) {
                    sl_tri_fill_only<DT>(vt, x, sc_l, A, idx - lo);
                    v_new[idx - lo] = best_q;
                    if (argmax) argmax[idx - lo] = best_a;
                    double v_old = vt.table[idx * vt.ncols];         // stats[1] is policy-mode only
                    if (M.m.value.negate) v_old = v_old * -1.0;
                    lmax = fmax(lmax, fabs(best_q - v_old));
                }
            }
            SL_BM_SYNC();
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
        lsum += __shfl_xor(lsum, o, 64);
    }
    if (lane == 0) { red_max[wave] = lmax; red_sum[wave] = lsum; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SL_BM_WAVES; ++w) { lmax = fmax(lmax, red_max[w]); lsum += red_sum[w]; }
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        atomicAdd(&stats[1], lsum);
    }
}

#undef SL_BM_SYNC

// ---------------------------------------------------------------------------------------------
// Policy-evaluation sweep on the matrix cores (V <- r(x, pi(x)) + gamma V(f(x, pi(x))))
// ---------------------------------------------------------------------------------------------
// The policies of the value-iteration loop are piecewise constant (the greedy table over a finite
// action set), so the 64 consecutive cells of a wavefront - one row of the last grid axis, hence
// one common product P[j] = sigma^2 prod_k T_k[i_k][j] of the leading axes' tables, kept in LDS -
// take only a few distinct actions.  Up to G = 16 / dout of them share one GEMM
//   mean[64 cells x (g, dd)] = T_last[cells x n] . B[n x (g, dd)],  B = P[j] E_j(u_g) alpha'[j, dd]:
// the A operand is the last axis' table entry itself, the B element costs its lane one exponential
// per slab, hidden under the four MFMAs.  Rows with more than SL_BP_MAXG distinct actions (smooth
// policies) take the scalar loop instead.
#define SL_BP_MAXG 12
template <int DT>
__global__ __launch_bounds__(64 * SL_BM_WAVES) void k_bellman_policy_mfma(
    const SlDevModel M, const SlGpDev gp, SlAux aux, SlBellmanPack pk, int64_t lo, int64_t hi,
    const double* __restrict__ pack, double* __restrict__ v_new, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_max[SL_BM_WAVES], red_sum[SL_BM_WAVES];
    const SlDims nd = sl_dims<DT, 1>(M);
    const int d = nd.d, p = nd.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    __shared__ SlTri vt_lds;
    sl_stage_tri(&vt_lds, &aux.tri[0]);
    const SlTri& vt = vt_lds;
    const int n_last = M.m.grid.num_points[d - 1];
    const int npad_max = pk.npad_max;
    double* p_l = smem + (size_t)wave * (npad_max + 64 * 16 + SL_BP_MAXG); // [n_pad] of the current head
    double* mean_l = p_l + npad_max;                                       // [64 cells][16 columns]
    double* dist_l = mean_l + 64 * 16;                                     // distinct actions of the row
    double lmax = 0.0, lsum = 0.0;
    // work item = one segment of at most 64 consecutive cells of one row of the last grid axis
    // (the cells of a segment share the leading-axis indices); rows that straddle [lo, hi) and
    // the ragged last segment of a row are masked
    const int64_t segs = (n_last + 63) / 64;
    const int64_t row_lo = lo / n_last, row_hi = (hi + n_last - 1) / n_last;
    const int64_t nitems = (row_hi - row_lo) * segs;
    const int64_t nsteps = (nitems + SL_BM_WAVES - 1) / SL_BM_WAVES;
    for (int64_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const int64_t item = step * SL_BM_WAVES + wave;
        if (item >= nitems) continue;
        const int64_t row = row_lo + item / segs;
        const int seg0 = (int)(item % segs) * 64;                          // first last-axis index
        const int64_t wbase = row * n_last + seg0;
        const int64_t idx = wbase + lane;
        const bool valid = seg0 + lane < n_last && idx >= lo && idx < hi;
        int64_t cidx = row * n_last + (seg0 + lane < n_last ? seg0 + lane : n_last - 1);
        cidx = cidx < lo ? lo : (cidx < hi ? cidx : hi - 1);
        if (__ballot(valid) == 0ull) continue;
        double x[SL_P], u[SL_M];
        sl_index_to_state(M.m.grid, M.gf, d, cidx, x);
        sl_policy_any<true>(M, nd, aux.tri, cidx, x, u);
        // distinct actions of the row and this lane's group
        int gid = -1, ng = 0;
        uint64_t remaining = __ballot(valid);
        while (remaining && ng < SL_BP_MAXG) {
            const int leader = __ffsll((unsigned long long)remaining) - 1;
            const double ul = __shfl(u[0], leader, 64);
            const bool mine = valid && fabs(u[0] - ul) <= 1e-14 * (1.0 + fabs(ul));
            const uint64_t same = __ballot(mine) & remaining;
            if ((same >> lane) & 1ull) gid = ng;
            if (lane == 0) dist_l[ng] = ul;
            remaining &= ~same;
            ++ng;
        }
        const bool use_gemm = remaining == 0ull;
        // leading-axis indices of this wavefront's grid row
        int64_t ijk[SL_D];
        sl_unravel(M.m.grid, M.gf, d, wbase, ijk);
        double mean_state[SL_D];                         // posterior mean per state dimension
#pragma unroll
        for (int k = 0; k < SL_D; ++k) mean_state[k] = 0.0;
        for (int h = 0; h < pk.nheads; ++h) {            // FunctionStack: one pass per head
        const SlGpHeadDev& hd = gp.head[h];
        const int n_pad = hd.n_pad, nslab = n_pad / 4, dout = hd.dout;
        const double* tabs = pack + pk.tab0[h];
        const double inv_ls_u = hd.inv_ls[d];
        const double* __restrict__ xs_u = hd.xs + (size_t)d * n_pad;       // action inputs / l
        const int G = 16 / dout;                                           // actions per GEMM
        const int cg = lc / dout, cdd = lc - cg * dout;                    // this lane's B column
        const double* trow[SL_D];
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d - 1) trow[k] = tabs + pk.toff[h][k] + ijk[k] * n_pad;
        double mean[SL_D];
#pragma unroll
        for (int k = 0; k < SL_D; ++k) mean[k] = 0.0;
        if (use_gemm) {
            for (int j = lane; j < n_pad; j += 64) {
                double v = hd.variance;
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d - 1) v *= trow[k][j];
                p_l[j] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double* ap = tabs + pk.toff[h][d - 1] + (int64_t)lk * n_last;
            int acol[4];                                     // last-axis index of this lane per tile
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = seg0 + 16 * t + lc;
                acol[t] = c < n_last ? c : n_last - 1;
            }
            const size_t astep = (size_t)4 * n_last;
            for (int g0 = 0; g0 < ng; g0 += G) {
                const bool col_on = cg < G && g0 + cg < ng;
                const double ug = dist_l[col_on ? g0 + cg : 0] * inv_ls_u;
                const double cmask = col_on ? 1.0 : 0.0;
                const double* alp = hd.alpha + (size_t)lk * dout + cdd;    // + 4 dout per slab
                const double* xsp = xs_u + lk, *pp = p_l + lk;             // + 4 per slab
                auto b_elem = [&](int sl) -> double {
                    const double dlt = xsp[4 * sl] - ug;
                    return (pp[4 * sl] * cmask) * (exp(-0.5 * (dlt * dlt)) * alp[(size_t)sl * 4 * dout]);
                };
                sl_bd4 acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = (sl_bd4){0.0, 0.0, 0.0, 0.0};
                double a_cur[4], a_nxt[4], b_cur, b_nxt;
#pragma unroll
                for (int t = 0; t < 4; ++t) a_cur[t] = ap[acol[t]];
                b_cur = b_elem(0);
                for (int s = 0; s < nslab; s += 2) {
                    const int s1 = s + 1, s2 = s + 2 < nslab ? s + 2 : 0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) a_nxt[t] = ap[s1 * astep + acol[t]];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], b_cur, acc[t], 0, 0, 0);
                    b_nxt = b_elem(s1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) a_cur[t] = ap[s2 * astep + acol[t]];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_nxt[t], b_nxt, acc[t], 0, 0, 0);
                    b_cur = b_elem(s2);
                }
                // D tile: column (g, dd) = lane & 15, row (cell) = (lane >> 4) + 4 * reg
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double* dst = mean_l + (16 * t + lk) * 16 + lc;
                    dst[0] = acc[t].x;
                    dst[4 * 16] = acc[t].y;
                    dst[8 * 16] = acc[t].z;
                    dst[12 * 16] = acc[t].w;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (gid >= g0 && gid < g0 + G) {
#pragma unroll
                    for (int k = 0; k < SL_D; ++k)
                        if (k < dout) mean[k] = mean_l[lane * 16 + (gid - g0) * dout + k];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (valid) {
            // smooth policy: one exponential per (cell, training point)
            const double ug = u[0] * inv_ls_u;
            const double* tl = tabs + pk.toff[h][d - 1] + (seg0 + lane < n_last ? seg0 + lane : n_last - 1);
            for (int j = 0; j < hd.n; ++j) {
                double v = hd.variance * tl[(size_t)j * n_last];
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d - 1) v *= trow[k][j];
                const double dlt = xs_u[j] - ug;
                v *= sl_exp_nonpos(-0.5 * (dlt * dlt));
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < dout) mean[k] = fma(v, hd.alpha[(size_t)j * dout + k], mean[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < SL_D; ++k) {
            const int dd = k - hd.col0;
#pragma unroll
            for (int q = 0; q < SL_D; ++q)
                if (q == dd && q < dout) mean_state[k] = mean[q];
        }
        // (p_l and mean_l are rewritten by the next head only after this wavefront's reads)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }
        if (valid) {
            double prior[SL_D], nxt[SL_D];
            sl_append_action(nd, u, x);
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) if (k < d) nxt[k] = mean_state[k] + prior[k];
            const double r = sl_quadratic(M.m.reward, p, x);
            double v = sl_tri_value_fast<DT>(vt, nxt);
            if (M.m.value.negate) v = v * -1.0;
            const double tq = M.m.gamma * v;
            const double q = r + tq;
            v_new[idx - lo] = q;
            double v_old = vt.table[idx * vt.ncols];
            double v_int = sl_tri_value_fast<DT>(vt, x);
            if (M.m.value.negate) { v_old = v_old * -1.0; v_int = v_int * -1.0; }
            lmax = fmax(lmax, fabs(q - v_old));
            const double diff = q - v_int;
            lsum = fma(diff, diff, lsum);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
        lsum += __shfl_xor(lsum, o, 64);
    }
    if (lane == 0) { red_max[wave] = lmax; red_sum[wave] = lsum; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SL_BM_WAVES; ++w) { lmax = fmax(lmax, red_max[w]); lsum += red_sum[w]; }
        atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]),
                  (unsigned long long)__double_as_longlong(lmax));
        atomicAdd(&stats[1], lsum);
    }
}

// Sets *done = 1 when the matrix-core path handled the sweep (one GP head whose outputs span the
// state, at most 96 (action, output) columns); otherwise the caller runs the VALU kernel.
static int bellman_mfma(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, double* d_v_new,
                        int32_t* d_argmax, double* d_q, double* d_stats, int* done) {
    *done = 0;
    const SlDevModel& M = ctx->h_model;
    if (ctx->env.bellman_mfma == 0) return SL_OK;
    const int nheads = ctx->h_gp.nheads;
    if (M.m.policy.m != 1 || nheads < 1 || nheads > SL_BM_HEADS) return SL_OK;
    const int variant = sl_dim_variant_of(M);
    if (variant != 4 && variant != 2) return SL_OK;        // compiled for 2 and 4 state dimensions
    const SlGpHeadHost& hh = ctx->gp_heads[0];
    const int d = M.m.grid.d;
    SlBellmanPack pk;
    memset(&pk, 0, sizeof(pk));
    const bool policy_mode = n_actions == 0;
    if (!policy_mode) {
        // the 4x4x4 kernel (sl_bellman4.hip) takes the sweeps it is built for
        int done4 = 0;
        const int rc4 = sl_bellman4_launch(ctx, lo, hi, n_actions, d_v_new, d_argmax, d_q, d_stats, &done4);
        if (rc4 != SL_OK) return rc4;
        if (done4) { *done = 1; return SL_OK; }
    }
    if (policy_mode) {
        // worthwhile for piecewise-constant policies; one shared-input head
        const int pkind = M.m.policy.kind;
        if (pkind != SL_POLICY_TRI && pkind != SL_POLICY_TABLE && pkind != SL_POLICY_CONST) return SL_OK;
        // few distinct actions, last axis a multiple of 64 cells: the 4x4x4 kernel (sl_bellman4.hip)
        int done4 = 0;
        const int rc4 = sl_bellman4_policy_launch(ctx, lo, hi, d_v_new, d_stats, &done4);
        if (rc4 != SL_OK) return rc4;
        if (done4) { *done = 1; return SL_OK; }
        for (int h = 0; h < nheads; ++h)
            if (ctx->gp_heads[h].dout > SL_D) return SL_OK;
    }
    // one head with D outputs (1, 3 or 6 column blocks), or a FunctionStack of 2 (d = 2) / 4 (d = 4)
    // single-output heads with one column block each
    int ncb = 0, n_pad_max = 0;
    for (int h = 0; h < nheads; ++h) {
        const int c = (n_actions * ctx->gp_heads[h].dout + 15) / 16;
        ncb = c > ncb ? c : ncb;
        n_pad_max = ctx->gp_heads[h].n_pad > n_pad_max ? ctx->gp_heads[h].n_pad : n_pad_max;
    }
    if (ncb > 6) return SL_OK;
    if (!policy_mode && nheads > 1 && (ncb > 1 || nheads != d)) return SL_OK;
    const int ncb_t = policy_mode ? 0 : (ncb <= 1 ? 1 : (ncb <= 3 ? 3 : 6));
    pk.ncb = ncb_t;
    pk.nheads = nheads;
    pk.npad_max = n_pad_max;
    pk.rowlen = 16 * ncb_t * nheads + 1;          // odd: the epilogue reads one row per lane
    int64_t cursor = 0;
    for (int h = 0; h < nheads; ++h) {
        const int n_pad = ctx->gp_heads[h].n_pad;
        pk.boff[h] = cursor;
        cursor += (int64_t)(n_pad / 4) * ncb_t * 64;
        pk.tab0[h] = cursor;
        int64_t toff = 0;
        for (int k = 0; k < d; ++k) {
            pk.toff[h][k] = toff;
            toff += (int64_t)M.m.grid.num_points[k] * n_pad;
        }
        if (toff + 4 * (int64_t)n_pad * M.m.grid.num_points[d - 1] > 0x0fffffffll) return SL_OK;
        cursor += toff;
    }
    const size_t lds = policy_mode
        ? sizeof(double) * (size_t)SL_BM_WAVES * (n_pad_max + 64 * 16 + SL_BP_MAXG)
        : sizeof(double) * (size_t)SL_BM_WAVES * SL_BM_SUB_OF(ncb_t * nheads) * pk.rowlen;
    if (lds + sizeof(SlTri) + 512 > 160 * 1024) return SL_OK;
    const size_t need = sizeof(double) * (size_t)cursor;
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    double* pack = reinterpret_cast<double*>(ctx->d_scratch);
    hipLaunchKernelGGL(k_bellman_pack, dim3(512), dim3(256), 0, ctx->stream, ctx->h_model, ctx->h_gp,
                       pk, n_actions, ctx->d_actions, pack);
    SL_HIP_CHECK(ctx, hipGetLastError());
    const int64_t wg_cells = 16 * SL_BM_T * SL_BM_WAVES;
    const int64_t ntiles = (hi - lo + wg_cells - 1) / wg_cells;
    const int blocks = (int)(ntiles < ctx->num_cu ? ntiles : ctx->num_cu);
    SlAux aux{ctx->d_tri, ctx->d_net};
    if (policy_mode) {
#define SL_BP_LAUNCH(D_)                                                                          \
        do {                                                                                      \
            auto kern = k_bellman_policy_mfma<D_>;                                                \
            SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),            \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                                  (int)lds));                                     \
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SL_BM_WAVES), lds, ctx->stream,      \
                               ctx->h_model, ctx->h_gp, aux, pk, lo, hi, pack, d_v_new, d_stats); \
        } while (0)
        if (variant == 4) SL_BP_LAUNCH(4); else SL_BP_LAUNCH(2);
#undef SL_BP_LAUNCH
        SL_HIP_CHECK(ctx, hipGetLastError());
        sl_note_kernel(ctx, false, "k_bellman_policy_mfma<d=%d>", variant == 4 ? 4 : 2);
        *done = 1;
        return SL_OK;
    }
    const int bm_flags = sl_diag_flags("SL_BM_FLAGS");  // (development builds only: 0 in the shipped library)
    SlSuccDev fill;
    memset(&fill, 0, sizeof(fill));
    if (ctx->succ.filling && !(bm_flags & 2)) {
        fill = sl_succ_view(ctx);
        ctx->succ.filled = true;
    }
#define SL_BM_LAUNCH(D_, N_, H_)                                                                  \
    do {                                                                                          \
        auto kern = k_bellman_mfma<D_, N_, H_>;                                                   \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)lds));                                         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SL_BM_WAVES), lds, ctx->stream,          \
                           ctx->h_model, ctx->h_gp, aux, pk, lo, hi, n_actions, ctx->d_actions,   \
                           pack, d_v_new, d_argmax, d_q, d_stats, bm_flags, fill);                \
    } while (0)
#define SL_BM_DIMS(N_)                                  \
    do {                                                \
        if (variant == 4) SL_BM_LAUNCH(4, N_, 1);       \
        else SL_BM_LAUNCH(2, N_, 1);                    \
    } while (0)
    if (nheads > 1) {
        if (variant == 4) SL_BM_LAUNCH(4, 1, 4);
        else SL_BM_LAUNCH(2, 1, 2);
    } else if (ncb_t == 1) SL_BM_DIMS(1);
    else if (ncb_t == 3) SL_BM_DIMS(3);
    else SL_BM_DIMS(6);
#undef SL_BM_DIMS
#undef SL_BM_LAUNCH
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_bellman_mfma<d=%d, column blocks=%d, heads=%d>", variant == 4 ? 4 : 2,
                   nheads > 1 ? 1 : ncb_t, nheads);
    *done = 1;
    return SL_OK;
}

static int bellman_sweep_uncached(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions,
                                  const double* h_actions, double* d_v_new, int32_t* d_argmax,
                                  double* d_q, double* d_stats, bool is_gp, bool other_kernels);

extern "C" int sl_bellman_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions,
                                const double* h_actions, double* d_v_new, int32_t* d_argmax,
                                double* d_q, double* d_stats) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_bellman_sweep: NULL context");
    SlTimed timed(ctx, 2);
    if (!ctx->model_set) return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: call sl_model_set first");
    if (!ctx->h_tri[0].set)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: value table (sl_tri_set slot 0) not set");
    const SlDevModel& M = ctx->h_model;
    if (M.m.value.kind != SL_V_TRI)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: the value function must be a Triangulation");
    if (M.m.reward.kind != SL_V_QUADRATIC)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: reward must be a QuadraticFunction");
    if (ctx->h_tri[0].grid.d != M.m.grid.d)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: value grid / model grid mismatch");
    for (int k = 0; k < M.m.grid.d; ++k)
        if (ctx->h_tri[0].grid.num_points[k] != M.m.grid.num_points[k])
            return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: value grid / model grid mismatch");
    if (lo < 0 || hi < lo || hi > M.gf.nindex || !d_v_new || !d_stats)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: bad range or NULL output");
    if (n_actions < 0 || n_actions > SL_MAX_ACTIONS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: %d actions (max %d)", n_actions,
                       SL_MAX_ACTIONS);
    if (n_actions > 0 && !h_actions)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: NULL action list");
    if (M.m.policy.kind == SL_POLICY_TRI && !ctx->h_tri[1].set)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: policy table not set");
    const bool is_gp = M.m.dynamics.kind == SL_DYN_GP;
    if (is_gp && ctx->h_gp.nheads < 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bellman_sweep: GP dynamics without heads");
    // heads with a sum-of-products kernel: the matrix-core sweeps generate RBF values (state and
    // action factors); such models take k_bellman's one-(x, u)-at-a-time path
    bool other_kernels = false;
    if (is_gp)
        for (int h = 0; h < ctx->h_gp.nheads; ++h) other_kernels = other_kernels || ctx->gp_heads[h].d_kernel;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipMemsetAsync(d_stats, 0, 2 * sizeof(double), ctx->stream));
    ctx->last_kernel[0] = 0;
    if (hi == lo) return SL_OK;
    // policy evaluation with a network policy: one action per vertex first (a max sweep ignores the policy)
    SlPolicyTableScope network_policy(ctx, n_actions == 0 ? lo : 0, n_actions == 0 ? hi : 0, nullptr);
    if (network_policy.rc) return network_policy.rc;
    // The successors of (vertex, action) do not depend on the value table
    // (reinforcement_learning.py:89-104): a sweep over a range / action set / dynamics that an
    // earlier max sweep located is served from the successor cache (sl_succ.hip); otherwise a max
    // sweep fills it on its way (the kernels that can: k_bellman_lookup, k_bellman).
    {
        int done = 0;
        int rc = sl_succ_sweep(ctx, lo, hi, n_actions, h_actions, d_v_new, d_argmax, d_q, d_stats, &done);
        if (rc) return rc;
        if (done) return SL_OK;
    }
    ctx->succ.filling = ctx->succ.filled = false;
    if (n_actions > 0) ctx->succ.filling = sl_succ_begin_fill(ctx, lo, hi, n_actions, h_actions).w != nullptr;
    const int rc_sweep = bellman_sweep_uncached(ctx, lo, hi, n_actions, h_actions, d_v_new, d_argmax, d_q,
                                                d_stats, is_gp, other_kernels);
    if (rc_sweep == SL_OK && ctx->succ.filling && ctx->succ.filled) sl_succ_commit(ctx);
    ctx->succ.filling = ctx->succ.filled = false;
    return rc_sweep;
}

static int bellman_sweep_uncached(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions,
                                  const double* h_actions, double* d_v_new, int32_t* d_argmax,
                                  double* d_q, double* d_stats, bool is_gp, bool other_kernels) {
    const SlDevModel& M = ctx->h_model;
    size_t lds = 0;
    int amax = 0;
    if (n_actions > 0) {
        SL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_actions, h_actions,
                                         sizeof(double) * n_actions * M.m.policy.m,
                                         hipMemcpyHostToDevice, ctx->stream));
        SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        amax = n_actions <= 3 ? 3 : (n_actions <= 9 ? 9 : SL_MAX_ACTIONS);
        if (is_gp && !other_kernels)
            for (int h = 0; h < ctx->h_gp.nheads; ++h)
                lds += sizeof(double) * (size_t)ctx->gp_heads[h].n_pad * amax;
    }
    if (is_gp && !other_kernels) {                // dense K_nm @ alpha contraction on the matrix cores
        int done = 0;
        int rc = bellman_mfma(ctx, lo, hi, n_actions, d_v_new, d_argmax, d_q, d_stats, &done);
        if (rc) return rc;
        if (done) return SL_OK;
    }
    if (lds > 150 * 1024)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "sl_bellman_sweep: action-factor table needs %zu "
                                                "bytes of LDS", lds);
    int64_t blocks64 = (hi - lo + SL_BLOCK - 1) / SL_BLOCK;
    const int cap = ctx->num_cu * 4;
    const int blocks = (int)(blocks64 < cap ? blocks64 : cap);
    SlAux aux{ctx->d_tri, ctx->d_net};
    const int variant = sl_dim_variant_of(M);
#define SL_BELLMAN(ACT, D_, M_)                                                                  \
    do {                                                                                         \
        auto kern = k_bellman<ACT, D_, M_>;                                                      \
        if (lds > 48 * 1024)                                                                     \
            SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),           \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                                  (int)lds));                                    \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(SL_BLOCK), lds, ctx->stream, ctx->h_model,   \
                           ctx->h_gp, aux, lo, hi, n_actions, ctx->d_actions, d_v_new, d_argmax, \
                           d_q, d_stats, fill);                                                  \
    } while (0)
#define SL_BELLMAN_DIMS(AM_)                                     \
    do {                                                        \
        if (variant == 4) SL_BELLMAN(AM_, 4, 1);                \
        else if (variant == 2) SL_BELLMAN(AM_, 2, 1);           \
        else if (variant == 1) SL_BELLMAN(AM_, 1, 1);           \
        else SL_BELLMAN(AM_, 0, 0);                             \
    } while (0)
    // (DT = 0, the runtime-dimension flavour, locates through sl_tri_eval: nothing to cache)
    SlSuccDev fill;
    memset(&fill, 0, sizeof(fill));
    if (ctx->succ.filling && variant != 0) {
        fill = sl_succ_view(ctx);
        ctx->succ.filled = true;
    }
    sl_note_kernel(ctx, false, "k_bellman<actions<=%d, d=%d>", amax, variant);
    if (amax == 3) SL_BELLMAN_DIMS(3);
    else if (amax == 9) SL_BELLMAN_DIMS(9);
    else if (amax == SL_MAX_ACTIONS) SL_BELLMAN_DIMS(SL_MAX_ACTIONS);
    else SL_BELLMAN_DIMS(0);
#undef SL_BELLMAN_DIMS
#undef SL_BELLMAN
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// =============================================================================================
// evaluation at explicit points
// =============================================================================================
__global__ __launch_bounds__(SL_BLOCK) void k_eval_simple(const SlDevModel M, SlAux aux, int what,
                                                          int64_t n, const double* __restrict__ points,
                                                          double* __restrict__ out) {
    const SlDims nd = sl_dims<0, 0>(M);
    const int d = nd.d;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P];
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) x[k] = points[i * d + k];
        if (what == SL_EVAL_VALUE) {
            out[i] = sl_value_any<SL_FULL>(M, d, aux, x);
        } else if (what == SL_EVAL_POLICY) {
            double u[SL_M];
            sl_policy_any<true>(M, nd, aux.tri, 0, x, u);
#pragma unroll
            for (int a = 0; a < SL_M; ++a) if (a < nd.m) out[i * nd.m + a] = u[a];
        } else {   // SL_EVAL_LV
            double lv[SL_D];
            sl_lv_any<SL_FULL>(M, d, aux, x, lv);
            const int cols = M.m.lipschitz.lv_cols;
#pragma unroll
            for (int k = 0; k < SL_D; ++k) if (k < cols) out[i * cols + k] = lv[k];
        }
    }
}

int sl_sweep_any(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                 const double* d_values, uint64_t* d_neg_bits, sl_sweep_result* d_result,
                 double* d_dbg, const double* d_points);

// rows 0 .. n-1 of a per-point action table through the policy's saturation
__global__ __launch_bounds__(SL_BLOCK) void k_policy_rows(const SlDevModel M, int64_t n, double* __restrict__ out) {
    const SlDims nd = sl_dims<0, 0>(M);
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P] = {}, u[SL_M];
        sl_policy_any<false>(M, nd, nullptr, i, x, u);
#pragma unroll
        for (int a = 0; a < SL_M; ++a) if (a < nd.m) out[i * nd.m + a] = u[a];
    }
}

extern "C" int sl_eval_points(sl_ctx* ctx, int what, int64_t n, const double* d_points,
                              double* d_out) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_eval_points: NULL context");
    if (!ctx->model_set) return sl_fail(ctx, SL_ERR_INVALID, "sl_eval_points: call sl_model_set first");
    if (n < 0 || !d_points || !d_out) return sl_fail(ctx, SL_ERR_INVALID, "sl_eval_points: bad argument");
    if (n == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // a network policy: one action per point first (V and L_v do not evaluate the policy)
    const bool needs_policy = what == SL_EVAL_POLICY || what == SL_EVAL_DYNAMICS || what == SL_EVAL_DECREASE;
    SlPolicyTableScope network_policy(ctx, 0, needs_policy ? n : 0, d_points);
    if (network_policy.rc) return network_policy.rc;
    const SlDevModel& M = ctx->h_model;
    if (what == SL_EVAL_POLICY && network_policy.swapped) {
        // the action table IS the answer (saturated like every policy, functions.py:349-354)
        int64_t b = (n + SL_BLOCK - 1) / SL_BLOCK;
        if (b > SL_MAX_GRID) b = SL_MAX_GRID;
        hipLaunchKernelGGL(k_policy_rows, dim3((unsigned)b), dim3(SL_BLOCK), 0, ctx->stream, ctx->h_model, n, d_out);
        SL_HIP_CHECK(ctx, hipGetLastError());
        return SL_OK;
    }
    if (what == SL_EVAL_VALUE || what == SL_EVAL_POLICY || what == SL_EVAL_LV) {
        if (what == SL_EVAL_POLICY && M.m.policy.kind == SL_POLICY_TABLE)
            return sl_fail(ctx, SL_ERR_UNSUPPORTED, "a per-vertex policy table cannot be evaluated "
                                                    "at arbitrary points");
        if (M.m.value.kind == SL_V_TRI && !ctx->h_tri[0].set && what != SL_EVAL_POLICY)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_eval_points: value table not set");
        if (M.m.value.kind == SL_V_NETWORK && !ctx->h_net.set && what != SL_EVAL_POLICY)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_eval_points: network not set");
        int64_t b = (n + SL_BLOCK - 1) / SL_BLOCK;
        if (b > SL_MAX_GRID) b = SL_MAX_GRID;
        SlAux aux{ctx->d_tri, ctx->d_net};
        hipLaunchKernelGGL(k_eval_simple, dim3((unsigned)b), dim3(SL_BLOCK), 0, ctx->stream,
                           ctx->h_model, aux, what, n, d_points, d_out);
        SL_HIP_CHECK(ctx, hipGetLastError());
        return SL_OK;
    }
    if (what != SL_EVAL_DYNAMICS && what != SL_EVAL_DECREASE)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_eval_points: unknown selector %d", what);
    // run the sweep kernels over the point list; d_out receives the per-point record
    // [decrease, threshold, mean[d], err[d]]
    const size_t need = sizeof(uint64_t) * (size_t)((n + 63) / 64) + sizeof(sl_sweep_result);
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    sl_sweep_result* res = reinterpret_cast<sl_sweep_result*>(ctx->d_scratch);
    uint64_t* bits = reinterpret_cast<uint64_t*>(res + 1);
    return sl_sweep_any(ctx, 0, n, nullptr, nullptr, bits, res, d_out, d_points);
}
