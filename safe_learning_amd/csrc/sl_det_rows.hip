// sl_det_rows.hip - the deterministic decrease check for LINEAR dynamics, a linear (saturated)
// policy and a quadratic Lyapunov function: the HBM-bound configuration (cart-pole 128^4 with the
// linearised model: 8 bytes + 2 bits of memory traffic per cell).
//
// k_det_sweep (sl_kernels.hip) evaluates every cell from scratch, ~165 unfused FP64 operations -
// its floor is the FP64 VALU, not the memory system.  Bit-exactness with the reference forbids
// re-association (one rounding per multiply and per add, left to right as in the oracle), but it
// does not forbid SHARING: along a row of the last grid axis only the last state coordinate
// changes, and every ordered sum of the check
//     u      = ((x0 K0 + x1 K1) + x2 K2) + x3 K3                 policy            functions.py:1567-1583
//     f_j    = (((x0 M_j0 + x1 M_j1) + x2 M_j2) + x3 M_j3) + u M_j4   dynamics
//     lin_j  = ((x0 P_0j + x1 P_1j) + x2 P_2j) + x3 P_3j         V(x) = sum_j lin_j x_j   :1534-1539
//     t_j    = ((x0 G_j0 + x1 G_j1) + x2 G_j2) + x3 G_j3         L_v(x) = |t|      lyapunov.py:246-263
// starts with a prefix over the leading coordinates that is THE SAME NUMBER, rounding included, for
// all cells of the row.  A thread therefore owns CPT = 8 consecutive cells of one row: the
// prefixes are computed once (their cost spread over 8 cells), each cell adds its own last terms
// in the reference's order.  ~95 operations per cell instead of ~165, results bit for bit those of
// k_det_sweep (tests/test_gpu_lyapunov.py::test_row_kernel_is_bit_identical, the 128^4 sample of
// tests/test_gpu_full_size.py).
//
// Memory: a thread reads its 8 values as four 16-byte loads (64 contiguous bytes per lane, 4 KiB
// per wavefront), one byte of the initial-set mask, and writes one byte of the decrease mask.
#include "sl_common.h"

typedef double sl_d2_t __attribute__((ext_vector_type(2)));

namespace detrows {
constexpr int CPT = 8;        // cells per thread (one byte of each bit mask)
}

template <int DT>
__global__ __launch_bounds__(SL_BLOCK) void k_det_rows(
    const SlDevModel M_arg, int64_t lo, int64_t hi, const uint8_t* __restrict__ init_bytes,
    const double* __restrict__ values, uint8_t* __restrict__ neg_bytes,
    sl_key* __restrict__ partials) {
    using namespace detrows;
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    constexpr int D = DT, P = DT + 1, L = DT - 1;       // L: the last state coordinate
    SlDevModel M = M_arg;
    // every constant in its own VGPR (same value in all lanes, opaque to the optimiser): as scalar
    // operands they outnumber the SGPRs and come back through v_readlane (k_det_sweep's finding)
#define SL_TO_VGPR(x) asm volatile("" : "+v"(x))
#pragma unroll
    for (int k = 0; k < D; ++k) {
        SL_TO_VGPR(M.m.grid.unit_maxes[k]);
        SL_TO_VGPR(M.m.grid.offset[k]);
        SL_TO_VGPR(M.m.policy.matrix[0][k]);
#pragma unroll
        for (int q = 0; q < P; ++q) SL_TO_VGPR(M.m.dynamics.matrix[k][q]);
#pragma unroll
        for (int q = 0; q < D; ++q) {
            SL_TO_VGPR(M.m.value.matrix[k][q]);
            SL_TO_VGPR(M.m.lipschitz.lv_matrix[k][q]);
        }
    }
    SL_TO_VGPR(M.m.policy.lower[0]);
    SL_TO_VGPR(M.m.policy.upper[0]);
    SL_TO_VGPR(M.m.lipschitz.lv_const);
    SL_TO_VGPR(M.m.lipschitz.lf_const);
    SL_TO_VGPR(M.m.lipschitz.tau);
#undef SL_TO_VGPR
    const sl_policy_desc& pol = M.m.policy;
    const sl_value_desc& val = M.m.value;
    const sl_lipschitz_desc& lip = M.m.lipschitz;
    const double (&dyn)[SL_MAX_STATE_DIM][SL_MAX_INPUT_DIM] = M.m.dynamics.matrix;
    const bool lv_linear = lip.lv_kind != SL_LIP_CONST;

    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    const int64_t span = (int64_t)SL_BLOCK * CPT;
    for (int64_t base = lo + (int64_t)blockIdx.x * span; base < hi; base += (int64_t)gridDim.x * span) {
        const int64_t i0 = base + (int64_t)threadIdx.x * CPT;        // first of this thread's cells
        if (i0 >= hi) continue;
        // ---- loads first: the arithmetic below hides their latency -------------------------------
        double v8[CPT];
        if (values) {
            const sl_d2_t* src = reinterpret_cast<const sl_d2_t*>(values + (i0 - lo));
#pragma unroll
            for (int t = 0; t < CPT / 2; ++t) { const sl_d2_t two = src[t]; v8[2 * t] = two.x; v8[2 * t + 1] = two.y; }
        }
        const unsigned init8 = init_bytes ? init_bytes[(i0 - lo) >> 3] : 0u;

        // ---- the row: leading coordinates and the prefixes they determine -------------------------
        int64_t ijk[SL_D];
        sl_unravel(M.m.grid, M.gf, D, i0, ijk);
        double x[P];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const double tt = (double)ijk[k] * M.m.grid.unit_maxes[k];               // functions.py:731
            x[k] = tt + M.m.grid.offset[k];
        }
        // ordered sums over the coordinates k < L (D = 1: there is no prefix, the first term is the
        // cell's own): acc = x0 m0; acc = acc + x1 m1; ...
        double u_pre = 0.0, f_pre[D], lin_pre[D], t_pre[D];
        if (L > 0) {
            u_pre = x[0] * pol.matrix[0][0];
#pragma unroll
            for (int k = 1; k < L; ++k) { const double t = x[k] * pol.matrix[0][k]; u_pre = u_pre + t; }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double a = x[0] * dyn[j][0], b = x[0] * val.matrix[0][j], c = x[0] * lip.lv_matrix[j][0];
#pragma unroll
                for (int k = 1; k < L; ++k) {
                    const double ta = x[k] * dyn[j][k];
                    a = a + ta;
                    const double tb = x[k] * val.matrix[k][j];
                    b = b + tb;
                    const double tc = x[k] * lip.lv_matrix[j][k];
                    c = c + tc;
                }
                f_pre[j] = a; lin_pre[j] = b; t_pre[j] = c;
            }
        }
        const double one_plus_lf = 1.0 + lip.lf_const;                               // lyapunov.py:287

        // ---- the 8 cells -----------------------------------------------------------------------
        const int last0 = (int)ijk[L];
        unsigned neg8 = 0u;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const double tl = (double)(last0 + c) * M.m.grid.unit_maxes[L];       // v_cvt_f64_i32
            const double xl = tl + M.m.grid.offset[L];
            x[L] = xl;
            // policy: u = (prefix) + x_L K_L, saturated                              functions.py:349-354
            const double tu = xl * pol.matrix[0][L];
            double u = L > 0 ? u_pre + tu : tu;
            if (pol.saturate) {
                u = (u > pol.lower[0]) ? u : pol.lower[0];
                u = (u < pol.upper[0]) ? u : pol.upper[0];
            }
            x[D] = u;
            // dynamics f = [x, u] M^T, V(x), L_v(x)
            double f[D], vx = 0.0, l1 = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double ta = xl * dyn[j][L];
                const double a = L > 0 ? f_pre[j] + ta : ta;
                const double tb = u * dyn[j][D];
                f[j] = a + tb;
                const double tc = xl * val.matrix[L][j];
                const double lin = L > 0 ? lin_pre[j] + tc : tc;
                const double q = lin * x[j];
                vx = (j == 0) ? q : (vx + q);
                if (lv_linear) {
                    const double td = xl * lip.lv_matrix[j][L];
                    const double tj = L > 0 ? t_pre[j] + td : td;
                    // ABS_LINEAR: |t_j| per column, L1 over the columns in sl_threshold;
                    // NORM_LINEAR: the same sum formed in sl_lv (one column)
                    l1 = (j == 0) ? fabs(tj) : (l1 + fabs(tj));
                }
            }
            if (val.negate) vx = vx * -1.0;
            if (!lv_linear) l1 = lip.lv_const;                   // the scalar is used as it is
            // V(f): the full quadratic form                                          functions.py:1534-1539
            double vn = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double lin = f[0] * val.matrix[0][j];
#pragma unroll
                for (int k = 1; k < D; ++k) { const double t = f[k] * val.matrix[k][j]; lin = lin + t; }
                const double q = lin * f[j];
                vn = (j == 0) ? q : (vn + q);
            }
            if (val.negate) vn = vn * -1.0;
            // decrease and threshold                                                lyapunov.py:282-288, 376
            const double dv = vn - vx;
            const double decrease = dv + 0.0;
            const double thr0 = (-l1) * one_plus_lf;
            const double threshold = thr0 * lip.tau;
            const bool valid = i0 + c < hi;
            const bool negative = (decrease < threshold) && valid;
            neg8 |= negative ? (1u << c) : 0u;
            const bool ok = negative || ((init8 >> c) & 1u);
            // cells are visited in ascending index order (inside a thread and from one iteration
            // to the next): a strict "<" on the value bits keeps the lexicographic minimum
            if (valid && !ok) {
                const uint64_t vb = sl_vbits(values ? v8[c] : vx);
                if (vb < best_v || best_i == INT64_MAX) { best_v = vb; best_i = i0 + c; }   // (a NaN key is all ones)
            }
        }
        neg_bytes[(i0 - lo) >> 3] = (uint8_t)neg8;
    }
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (threadIdx.x == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// =============================================================================================
// host side
// =============================================================================================
// Models and ranges the row kernel takes: closed-form linear policy (one action), linear dynamics,
// quadratic V, L_v a scalar or |x G^T| (per column or as one norm), scalar L_f, 1..4 state
// dimensions, a last axis of whole bytes (multiple of 8 cells), a range of whole mask words.
bool sl_det_rows_supports(const SlDevModel& M, int64_t lo, int64_t hi) {
    const int d = M.m.grid.d;
    if (d < 1 || d > 4 || M.m.policy.m != 1) return false;
    if (M.m.policy.kind != SL_POLICY_LINEAR || M.m.dynamics.kind != SL_DYN_LINEAR) return false;
    if (M.m.value.kind != SL_V_QUADRATIC) return false;
    const int lv = M.m.lipschitz.lv_kind;
    if (lv != SL_LIP_CONST && lv != SL_LIP_ABS_LINEAR && lv != SL_LIP_NORM_LINEAR) return false;
    if (M.m.lipschitz.lf_kind != SL_LF_CONST) return false;
    if (M.m.grid.num_points[d - 1] % detrows::CPT) return false;
    return (lo % 64) == 0 && hi > lo && ((hi - lo) % detrows::CPT) == 0;
}

int sl_det_rows_launch(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                       const double* d_values, uint64_t* d_neg_bits, int* nblocks) {
    using namespace detrows;
    const SlDevModel& M = ctx->h_model;
    const int64_t span = (int64_t)SL_BLOCK * CPT;
    int64_t blocks = (hi - lo + span - 1) / span;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks > SL_MAX_GRID) blocks = SL_MAX_GRID;
    *nblocks = (int)blocks;
    // the last mask word of a range that is not a multiple of 64 cells: whole bytes are written
    // up to the range's end, the remaining bytes of that word are cleared first
    if ((hi - lo) & 63)
        SL_HIP_CHECK(ctx, hipMemsetAsync(d_neg_bits + ((hi - lo) >> 6), 0, sizeof(uint64_t), ctx->stream));
    const uint8_t* init_bytes = reinterpret_cast<const uint8_t*>(d_init_bits);
    uint8_t* neg_bytes = reinterpret_cast<uint8_t*>(d_neg_bits);
#define SL_ROWS(D_)                                                                                \
    hipLaunchKernelGGL(k_det_rows<D_>, dim3((unsigned)blocks), dim3(SL_BLOCK), 0, ctx->stream, M,  \
                       lo, hi, init_bytes, d_values, neg_bytes, ctx->d_partials)
    switch (M.m.grid.d) {
        case 1: SL_ROWS(1); break;
        case 2: SL_ROWS(2); break;
        case 3: SL_ROWS(3); break;
        default: SL_ROWS(4); break;
    }
#undef SL_ROWS
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_det_rows<d=%d> (8 cells of a row per thread)", M.m.grid.d);
    return SL_OK;
}
