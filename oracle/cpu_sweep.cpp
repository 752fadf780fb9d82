// oracle/cpu_sweep.cpp - TEST / MEASUREMENT INFRASTRUCTURE, not product code.
//
// The CPU baseline of SURVEY 8d at host strength: the reference's verification loop
// (safe_learning/lyapunov.py:517-529: batches of config.gp_batch_size cells in ascending-V order fed
// through  policy -> GP posterior -> V(mean) - V(x) + L_v(mean) . err < threshold)  restated in C++
// with OpenMP over all the cores bench.py finds, for the workloads of BASELINE.json's GP configs
// (saturated linear policy, one shared-kernel RBF GPRCached with a linear prior mean, quadratic V,
// L_v = |x (2P)|, scalar L_f).  Only bench.py's cpu_baseline leg and tests/ may load it; nothing
// under safe_learning_amd/ does (tests/test_abi.py::test_product_never_imports_the_oracle).
//
// What it follows, per cell (the NumPy oracle oracle/np_functions.py, np_lyapunov.py restates the
// same lines and is the CHECKER of this file, tests/test_cpu_sweep.py):
//   x      = ijk * unit_maxes + offset                         functions.py:728-731
//   u      = clip(x K^T, lower, upper)                         functions.py:349-354, 1567-1583
//   Kx     = variance exp(-1/2 sum_q ((X_jq - z_q) / l_q)^2)   functions.py:438 (gpflow 0.4.0 RBF)
//   a      = L^-1 Kx   (forward substitution with the Cholesky factor, tf.matrix_triangular_solve)
//                                                              functions.py:441
//   mean   = a^T alpha + m(z),  var = variance - sum a^2       functions.py:442, 450-451
//   err    = beta sqrt(var)                                    functions.py:514
//   decrease  = V(mean) - V(x) + sum_j L_v(mean)_j err_j       lyapunov.py:344-352, 376
//   threshold = -|L_v(x)|_1 (1 + L_f) tau                      lyapunov.py:282-288
//   negative  = decrease < threshold                           lyapunov.py:441
// Cost per cell: n^2 + n(4p + 2) + 2nD + 2n flops (1.081 MFLOP at n = 1024, p = 5, D = 4),
// almost all of it in the triangular solve.
//
// Shape: a thread owns a tile of 32 cells (four 8-lane vectors: AVX-512 where the host has it, the
// compiler splits the vectors otherwise); the tile's Kx / a panel [n][32] (256 KB at n = 1024) stays
// in the core's L2, the factor is streamed from the shared L3; the solve is register-blocked four
// rows at a time (16 accumulators).  exp is a degree-13 polynomial after range reduction (x <= 0).
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

typedef double v8 __attribute__((vector_size(64)));
typedef long long i8 __attribute__((vector_size(64)));

#define SLC_MAXD 4
#define SLC_MAXP 5
constexpr int TC = 32;                    // cells per tile
constexpr int NV = TC / 8;                // vectors per tile row

extern "C" {
struct sl_cpu_model {
    int32_t d, n;                          // state dimension (action dimension 1), training points
    int64_t num_points[SLC_MAXD];
    double offset[SLC_MAXD], unit_maxes[SLC_MAXD];
    double K[SLC_MAXD];                    // policy row: u = x K^T
    int32_t saturate, reserved;
    double lower, upper;
    const double* X;                       // [n][p] training inputs
    const double* L;                       // [n][n] lower Cholesky factor of K + sigma_n^2 I
    const double* alpha;                   // [n][d] L^-1 (Y - m(X))
    double variance, lengthscales[SLC_MAXP], beta;
    double prior[SLC_MAXD][SLC_MAXP];      // m(z) = prior z
    double P[SLC_MAXD][SLC_MAXD];          // V(x) = x P x^T
    double G[SLC_MAXD][SLC_MAXD];          // L_v(x) = |x G^T|
    double lf, tau;
};
}

static inline v8 splat(double v) { return (v8){v, v, v, v, v, v, v, v}; }

// exp(x) for x <= 0 (deep underflow flushes to 0)
static inline v8 exp_nonpos(v8 x) {
    const v8 lo = splat(-700.0);
    x = x < lo ? lo : x;
    const v8 magic = splat(6755399441055744.0);                 // 1.5 * 2^52: rint by addition
    v8 k = x * splat(1.4426950408889634) + magic;
    const i8 ki = (i8)k;                                        // low bits = the integer
    k = k - magic;
    v8 r = x - k * splat(6.93147180369123816490e-01);
    r = r - k * splat(1.90821492927058770002e-10);
    v8 q = splat(1.6059043836821613e-10);
    q = q * r + splat(2.08767569878681e-09);
    q = q * r + splat(2.505210838544172e-08);
    q = q * r + splat(2.755731922398589e-07);
    q = q * r + splat(2.7557319223985893e-06);
    q = q * r + splat(2.48015873015873e-05);
    q = q * r + splat(1.984126984126984e-04);
    q = q * r + splat(1.3888888888888889e-03);
    q = q * r + splat(8.333333333333333e-03);
    q = q * r + splat(4.1666666666666664e-02);
    q = q * r + splat(1.6666666666666666e-01);
    q = q * r + splat(0.5);
    q = q * r + splat(1.0);
    q = q * r + splat(1.0);
    // q * 2^k: add k to the exponent field (k >= -1010: the result stays normal)
    const i8 bits = (i8)q + (ki << 52);
    return (v8)bits;
}

static void tile(const sl_cpu_model& m, const double* __restrict__ xs /* [n][p] X / l */,
                 const int64_t* idx, int count, double* __restrict__ a /* [n][TC] */, uint8_t* negative,
                 double* records) {
    const int d = m.d, p = d + 1, n = m.n;
    alignas(64) double x[SLC_MAXP][TC], zs[SLC_MAXP][TC];
    for (int c = 0; c < TC; ++c) {
        int64_t r = idx[c < count ? c : count - 1];
        double u = 0.0;
        for (int k = d - 1; k >= 0; --k) {
            const int64_t q = r / m.num_points[k];
            x[k][c] = (double)(r - q * m.num_points[k]) * m.unit_maxes[k] + m.offset[k];
            r = q;
        }
        for (int k = 0; k < d; ++k) u += x[k][c] * m.K[k];
        if (m.saturate) u = fmin(fmax(u, m.lower), m.upper);
        x[d][c] = u;
        for (int q = 0; q < p; ++q) zs[q][c] = x[q][c] / m.lengthscales[q];
    }
    // kernel row: Kx[j][c]
    const v8 var = splat(m.variance);
    for (int j = 0; j < n; ++j) {
        const double* xj = xs + (size_t)j * p;
        for (int v = 0; v < NV; ++v) {
            v8 r2 = splat(0.0);
            for (int q = 0; q < p; ++q) {
                const v8 dl = splat(xj[q]) - *(const v8*)&zs[q][8 * v];
                r2 += dl * dl;
            }
            *(v8*)&a[(size_t)j * TC + 8 * v] = var * exp_nonpos(splat(-0.5) * r2);
        }
    }
    // forward substitution a = L^-1 Kx, four rows at a time; sum of squares and a^T alpha on the way
    v8 ss[NV], mean[SLC_MAXD][NV];
    for (int v = 0; v < NV; ++v) {
        ss[v] = splat(0.0);
        for (int k = 0; k < SLC_MAXD; ++k) mean[k][v] = splat(0.0);
    }
    for (int i0 = 0; i0 < n; i0 += 4) {
        const int rows = n - i0 < 4 ? n - i0 : 4;
        v8 acc[4][NV];
        for (int r = 0; r < 4; ++r)
            for (int v = 0; v < NV; ++v)
                acc[r][v] = r < rows ? *(const v8*)&a[(size_t)(i0 + r) * TC + 8 * v] : splat(0.0);
        const double* l0 = m.L + (size_t)i0 * n;
        const double* l1 = m.L + (size_t)(i0 + (rows > 1 ? 1 : 0)) * n;
        const double* l2 = m.L + (size_t)(i0 + (rows > 2 ? 2 : 0)) * n;
        const double* l3 = m.L + (size_t)(i0 + (rows > 3 ? 3 : 0)) * n;
        for (int j = 0; j < i0; ++j) {
            const v8 c0 = splat(l0[j]), c1 = splat(l1[j]), c2 = splat(l2[j]), c3 = splat(l3[j]);
            for (int v = 0; v < NV; ++v) {
                const v8 aj = *(const v8*)&a[(size_t)j * TC + 8 * v];
                acc[0][v] -= c0 * aj;
                acc[1][v] -= c1 * aj;
                acc[2][v] -= c2 * aj;
                acc[3][v] -= c3 * aj;
            }
        }
        for (int r = 0; r < rows; ++r) {
            const double* lr = m.L + (size_t)(i0 + r) * n;
            for (int r2 = 0; r2 < r; ++r2) {
                const v8 c = splat(lr[i0 + r2]);
                for (int v = 0; v < NV; ++v) acc[r][v] -= c * *(const v8*)&a[(size_t)(i0 + r2) * TC + 8 * v];
            }
            const v8 diag = splat(lr[i0 + r]);
            const double* al = m.alpha + (size_t)(i0 + r) * d;
            for (int v = 0; v < NV; ++v) {
                const v8 ai = acc[r][v] / diag;
                *(v8*)&a[(size_t)(i0 + r) * TC + 8 * v] = ai;
                ss[v] += ai * ai;
                for (int k = 0; k < d; ++k) mean[k][v] += ai * splat(al[k]);
            }
        }
    }
    // per-cell check
    for (int c = 0; c < count; ++c) {
        const int v = c >> 3, l = c & 7;
        double mu[SLC_MAXD], err[SLC_MAXD], lvx = 0.0, vx = 0.0, vm = 0.0, bound = 0.0;
        const double fvar = m.variance - ss[v][l];
        for (int k = 0; k < d; ++k) {
            double prior = 0.0;
            for (int q = 0; q < p; ++q) prior += m.prior[k][q] * x[q][c];
            mu[k] = mean[k][v][l] + prior;
            err[k] = m.beta * sqrt(fvar);
        }
        for (int i = 0; i < d; ++i) {
            double rx = 0.0, rm = 0.0, gx = 0.0, gm = 0.0;
            for (int k = 0; k < d; ++k) {
                rx += x[k][c] * m.P[k][i];
                rm += mu[k] * m.P[k][i];
                gx += x[k][c] * m.G[i][k];
                gm += mu[k] * m.G[i][k];
            }
            vx += rx * x[i][c];
            vm += rm * mu[i];
            lvx += fabs(gx);
            bound += fabs(gm) * err[i];
        }
        const double decrease = vm - vx + bound;
        const double threshold = -lvx * (1.0 + m.lf) * m.tau;
        if (negative) negative[c] = decrease < threshold ? 1 : 0;
        if (records) {
            double* rec = records + (size_t)c * (2 + 2 * d);
            rec[0] = decrease;
            rec[1] = threshold;
            for (int k = 0; k < d; ++k) { rec[2 + k] = mu[k]; rec[2 + d + k] = err[k]; }
        }
    }
}

extern "C" {

// The decrease check at the cells idx[0 .. count) (flat C-order GridWorld indices), `threads` OpenMP
// threads (<= 0: all).  negative [count] (may be NULL), records [count][2 + 2d] = decrease,
// threshold, mean, err (may be NULL).  *seconds = wall time of the parallel region, *threads_used.
int sl_cpu_lyap_check(const sl_cpu_model* m, const int64_t* idx, int64_t count, int threads,
                      uint8_t* negative, double* records, double* seconds, int* threads_used) {
    if (!m || !idx || count < 0 || m->d < 1 || m->d > SLC_MAXD || m->n < 1) return -1;
    const int p = m->d + 1, n = m->n;
    double* xs = (double*)aligned_alloc(64, (((size_t)n * p * sizeof(double)) + 63) / 64 * 64);
    if (!xs) return -4;
    for (int j = 0; j < n; ++j)
        for (int q = 0; q < p; ++q) xs[(size_t)j * p + q] = m->X[(size_t)j * p + q] / m->lengthscales[q];
    if (threads <= 0) threads = omp_get_max_threads();
    const int64_t ntiles = (count + TC - 1) / TC;
    int failed = 0;
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(threads)
    {
        double* a = (double*)aligned_alloc(64, (size_t)n * TC * sizeof(double));
        if (!a) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp single
        if (threads_used) *threads_used = omp_get_num_threads();
        if (a) {
#pragma omp for schedule(dynamic, 4)
            for (int64_t t = 0; t < ntiles; ++t) {
                const int64_t c0 = t * TC;
                const int cnt = (int)(count - c0 < TC ? count - c0 : TC);
                tile(*m, xs, idx + c0, cnt, a, negative ? negative + c0 : nullptr,
                     records ? records + (size_t)c0 * (2 + 2 * m->d) : nullptr);
            }
            free(a);
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    free(xs);
    return failed ? -4 : 0;
}

int sl_cpu_max_threads(void) { return omp_get_max_threads(); }
}
