// sl_model.h - per-cell arithmetic of the Lyapunov sweep, shared by every kernel.
//
// Everything here is plain C++ on scalars (no HIP builtins), marked __host__ __device__ so
// that tests/hostsim can compile the very same functions with g++ and check them against
// the oracle without a GPU.  Arithmetic order is the canonical one defined by the oracle
// (oracle/np_functions.py: ordered_matmul / ordered_rowsum): left to right, one rounding per
// multiply and per add.  The translation unit is compiled with -ffp-contract=off, so no
// multiply-add below is ever fused; where a fused operation is wanted it is written fma().
//
// Reference formulas (upstream checkout):
//   state from index      safe_learning/functions.py:728-731
//   policy / linear dyn.  safe_learning/functions.py:1567-1583, saturation :349-354
//   quadratic V           safe_learning/functions.py:1534-1539
//   pendulum / cart-pole  examples/utilities.py:242-289 / 387-437
//   threshold             safe_learning/lyapunov.py:265-288
//   decrease + bound      safe_learning/lyapunov.py:324-376
#pragma once

#include <stdint.h>
#include <math.h>
#include <string.h>
#include "sl_hip.h"

#if defined(__HIPCC__)
#define SL_HD __host__ __device__ __forceinline__
#else
#define SL_HD inline
#endif

#define SL_D SL_MAX_STATE_DIM
#define SL_M SL_MAX_ACTION_DIM
#define SL_P SL_MAX_INPUT_DIM

// Device-side view of one auxiliary grid with a vertex table (Triangulation).
#define SL_TRI_CODES   64      // 2^(pairs of coordinates), d <= 4
#define SL_TRI_MAXCAND 12
struct SlTri {
    sl_grid_desc grid;
    int32_t nsimplex, project, ncols, set;
    int32_t simplices[SL_MAX_SIMPLICES][SL_D + 1];   // corner codes, bit k = dimension k
    double  hyper[SL_MAX_SIMPLICES][SL_D][SL_D];
    double  hyper_c[SL_MAX_SIMPLICES][SL_D];         // sum_k origin_s[k] * hyper[s][k][j] (sl_tri_finish)
    int64_t stride[SL_D];                            // flat-index stride per dimension
    const double* points;                            // concatenated linspace tables (device)
    int32_t points_off[SL_D];
    // 1: points[k][i] == (double)i * unit_maxes[k] + offset[k] bit for bit (np.linspace's own
    // formula, checked by sl_tri_finish) - the lookups then compute the points instead of loading
    int32_t affine_points, reserved;
    double  inv_unit[SL_D];                          // 1 / unit_maxes (first guesses only)
    // Point location shortcut (sl_tri_regions): the order of the unit-cell coordinates puts a point
    // into one of d! Kuhn regions; only the simplices that overlap that region can contain it.
    // region_code = one bit per coordinate pair (i < j): z_i < z_j.  ncand[code] == 0: no table.
    uint8_t ncand[SL_TRI_CODES];
    uint8_t cand[SL_TRI_CODES][SL_TRI_MAXCAND];
    // Second level (4-D only): inside a coordinate-order region the six signs of z_i + z_j - 1 cut
    // the candidates down to at most four.  fine[perm_index[code] * 64 + sum_code] = up to four
    // simplex indices, 0xFF = none / no entry (then the region's full list above is used).  Found
    // by sampling - complete up to slivers, and a miss only costs the full walk, never the result.
    int32_t has_fine, reserved2;
    uint8_t perm_index[SL_TRI_CODES];
    uint8_t fine[24 * 64][4];
    const double* table;                             // [nindex][ncols] (device)
};

// Device-side view of the LyapunovNetwork.
struct SlNet {
    int32_t nlayers, set;
    int32_t dims[SL_MAX_NN_LAYERS + 1];
    int32_t act[SL_MAX_NN_LAYERS];
    int32_t koff[SL_MAX_NN_LAYERS];                  // offset of layer kernel in `kernels`
    const double* kernels;                           // device, per layer [out][in]
    const double* kernels_t;                         // device, per layer [in][out] (same offsets)
    // zero-padded row-major copies for the MFMA kernels (sl_nn.hip): layer l is
    // [16 * nfb[l]][wstride[l]] at wpad + woff[l]
    const double* wpad;
    int32_t woff[SL_MAX_NN_LAYERS], wstride[SL_MAX_NN_LAYERS];
    int32_t nfb[SL_MAX_NN_LAYERS];                   // 16-wide blocks of output features
    int32_t nib[SL_MAX_NN_LAYERS];                   // 16-wide blocks of input features
    int32_t nslab[SL_MAX_NN_LAYERS];                 // 4-wide slabs of (padded) input features
    int32_t wtotal, reserved;
};

// ---------------------------------------------------------------------------------------------
// grid addressing
// ---------------------------------------------------------------------------------------------
struct SlGridFast {                 // derived from sl_grid_desc at sl_model_set time
    int32_t d, all_pow2;
    int32_t shift[SL_D];            // log2(num_points[k]) when all_pow2
    uint32_t num32[SL_D];
    int64_t nindex;
};

SL_HD void sl_unravel(const sl_grid_desc& g, const SlGridFast& f, int d, int64_t idx, int64_t* ijk) {
    if (f.all_pow2) {
        uint64_t r = (uint64_t)idx;
#pragma unroll
        for (int k = SL_D - 1; k >= 0; --k) {
            if (k < d) {
                ijk[k] = (int64_t)(r & ((1ull << f.shift[k]) - 1ull));
                r >>= f.shift[k];
            }
        }
    } else if (f.nindex <= 0xffffffffll) {
        uint32_t r = (uint32_t)idx;
#pragma unroll
        for (int k = SL_D - 1; k >= 0; --k) {
            if (k < d) {
                uint32_t q = r / f.num32[k];
                ijk[k] = (int64_t)(r - q * f.num32[k]);
                r = q;
            }
        }
    } else {
        int64_t r = idx;
#pragma unroll
        for (int k = SL_D - 1; k >= 0; --k) {
            if (k < d) {
                int64_t q = r / g.num_points[k];
                ijk[k] = r - q * g.num_points[k];
                r = q;
            }
        }
    }
}

// functions.py:731: ijk * unit_maxes + offset (multiply, then add)
SL_HD void sl_index_to_state(const sl_grid_desc& g, const SlGridFast& f, int d, int64_t idx,
                              double* x) {
    int64_t ijk[SL_D];
    sl_unravel(g, f, d, idx, ijk);
#pragma unroll
    for (int k = 0; k < SL_D; ++k) {
        if (k < d) {
            double t = (double)ijk[k] * g.unit_maxes[k];
            x[k] = t + g.offset[k];
        }
    }
}

// functions.py:612-638 (all_points): np.linspace makes the last point of every dimension exactly
// `upper`; the other points equal ijk * unit_maxes + offset.  Lyapunov.update_values evaluates V on
// these points (lyapunov.py:321), the verification loop on index_to_state (lyapunov.py:525).
SL_HD void sl_index_to_grid_point(const sl_grid_desc& g, const SlGridFast& f, int d, int64_t idx,
                                  double* x) {
    int64_t ijk[SL_D];
    sl_unravel(g, f, d, idx, ijk);
#pragma unroll
    for (int k = 0; k < SL_D; ++k) {
        if (k < d) {
            double t = (double)ijk[k] * g.unit_maxes[k];
            t = t + g.offset[k];
            x[k] = (ijk[k] == g.num_points[k] - 1) ? g.upper[k] : t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small ordered linear algebra: out[j] = ((z0*M[j][0] + z1*M[j][1]) + ...) rows = outputs
// ---------------------------------------------------------------------------------------------
template <int ROWS, int COLS>
SL_HD void sl_rows_dot(const double (&mat)[ROWS][COLS], int nrows, int ncols, const double* z,
                       double* out) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        if (j < nrows) {
            double acc = z[0] * mat[j][0];
#pragma unroll
            for (int i = 1; i < COLS; ++i) {
                if (i < ncols) {
                    double t = z[i] * mat[j][i];
                    acc = acc + t;
                }
            }
            out[j] = acc;
        }
    }
}

// functions.py:1537-1539: linear_form = z P ; sum_j linear_form_j * z_j
SL_HD double sl_quadratic(const sl_value_desc& v, int n, const double* z) {
    double total = 0.0;
#pragma unroll
    for (int j = 0; j < SL_P; ++j) {
        if (j < n) {
            double lin = z[0] * v.matrix[0][j];
#pragma unroll
            for (int i = 1; i < SL_P; ++i) {
                if (i < n) {
                    double t = z[i] * v.matrix[i][j];
                    lin = lin + t;
                }
            }
            double q = lin * z[j];
            total = (j == 0) ? q : (total + q);
        }
    }
    return v.negate ? (total * -1.0) : total;
}

// ---------------------------------------------------------------------------------------------
// policy
// ---------------------------------------------------------------------------------------------
SL_HD void sl_saturate(const sl_policy_desc& p, int m, double* u) {
    if (p.saturate) {
#pragma unroll
        for (int a = 0; a < SL_M; ++a) {
            if (a < m) {
                double t = u[a];
                t = (t > p.lower[a]) ? t : p.lower[a];      // tf.maximum(res, lower)
                t = (t < p.upper[a]) ? t : p.upper[a];      // tf.minimum(.., upper)
                u[a] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// analytic dynamics
// ---------------------------------------------------------------------------------------------
// sin and cos of one angle with a shared argument reduction: k = rint(x 2/pi), r = x - k pi/2 in
// three pieces (exact products for |k| < 2^20), fdlibm's kernel polynomials on |r| <= pi/4,
// quadrant fix-up.  About 35 instructions for the pair (the library calls take ~60 each) and
// within 1.5 ulp; the Euler integrators below call it 10 times per cell.
SL_HD void sl_sincos(double x, double* sn, double* cs) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(k, -1.57079632673412561417e+00, x);
    r = fma(k, -6.07710050630396597660e-11, r);
    r = fma(k, -2.02226624871116645580e-21, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double s_out = (q & 1) ? c : s;
    const double c_out = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_out : s_out;
    *cs = ((q + 1) & 2) ? -c_out : c_out;
}

// coef: [0]=dt/10 [1]=g/l [2]=inertia [3]=friction/inertia [4]=(friction>0)
SL_HD void sl_pendulum(const sl_dynamics_desc& f, const double* x, const double* u, double* nxt) {
    double th = x[0], om = x[1], act = u[0];
    if (f.normalize) { th = th * f.tx[0]; om = om * f.tx[1]; act = act * f.tu[0]; }
    const double dt = f.coef[0];
    for (int it = 0; it < 10; ++it) {
        double sth, cth;
        sl_sincos(th, &sth, &cth);
        double acc = f.coef[1] * sth;
        double t2 = act / f.coef[2];
        acc = acc + t2;
        if (f.coef[4] != 0.0) {
            double t3 = f.coef[3] * om;
            acc = acc - t3;
        }
        double dth = dt * om;
        double dom = dt * acc;
        th = th + dth;
        om = om + dom;
    }
    if (f.normalize) { th = th * f.tx_inv[0]; om = om * f.tx_inv[1]; }
    nxt[0] = th; nxt[1] = om;
}

// coef: [0]=dt/10 [1]=m [2]=M [3]=L [4]=b [5]=m*L [6]=((0.5*m)*g)*L [7]=(0.5*m)*L
//       [8]=b*(m+M) [9]=(m+M)*g
SL_HD void sl_cartpole(const sl_dynamics_desc& f, const double* x, const double* u, double* nxt) {
    double px = x[0], th = x[1], v = x[2], om = x[3], act = u[0];
    if (f.normalize) {
        px = px * f.tx[0]; th = th * f.tx[1]; v = v * f.tx[2]; om = om * f.tx[3];
        act = act * f.tu[0];
    }
    const double dt = f.coef[0], m = f.coef[1], M = f.coef[2], L = f.coef[3], b = f.coef[4];
    for (int it = 0; it < 10; ++it) {
        double s, c;
        sl_sincos(th, &s, &c);
        double s2 = 2.0 * (s * c);                   // sin(2 theta)
        double om2 = om * om;
        double det = m * (s * s);
        det = M + det;
        det = L * det;
        double t1 = (f.coef[5] * om2) * s;
        double t2 = (b * om) * c;
        double t3 = f.coef[6] * s2;
        double vd = act - t1;
        vd = vd - t2;
        vd = vd + t3;
        vd = vd * L;
        vd = vd / det;
        double w1 = act * c;
        double w2 = (f.coef[7] * om2) * s2;
        double w3 = (f.coef[8] * om) / f.coef[5];
        double w4 = f.coef[9] * s;
        double wd = w1 - w2;
        wd = wd - w3;
        wd = wd + w4;
        wd = wd / det;
        double dpx = dt * v, dth = dt * om, dv = dt * vd, dom = dt * wd;
        px = px + dpx; th = th + dth; v = v + dv; om = om + dom;
    }
    if (f.normalize) {
        px = px * f.tx_inv[0]; th = th * f.tx_inv[1]; v = v * f.tx_inv[2]; om = om * f.tx_inv[3];
    }
    nxt[0] = px; nxt[1] = th; nxt[2] = v; nxt[3] = om;
}

// ---------------------------------------------------------------------------------------------
// piecewise-linear interpolation on an auxiliary grid (functions.py:1103-1202, 1473-1499)
// ---------------------------------------------------------------------------------------------
// numpy.digitize(x, points) - 1 clipped to [0, n-2] (functions.py:771-773)
SL_HD int64_t sl_rectangle_1d(const double* pts, int64_t n, double offset, double unit, double x) {
    double g = (x - offset) / unit;
    int64_t i;
    if (!(g > 0.0)) i = 0; else if (g >= (double)(n - 1)) i = n - 1; else i = (int64_t)g;
    // digitize: number of points <= x ; fix the guess with the actual linspace values
    while (i + 1 < n && pts[i + 1] <= x) ++i;
    while (i > 0 && pts[i] > x) --i;
    int64_t cnt = (pts[i] <= x) ? (i + 1) : 0;    // bins[cnt-1] <= x < bins[cnt]
    if (x != x) cnt = n;                           // NaN sorts last in digitize
    int64_t r = cnt - 1;
    if (r < 0) r = 0;
    if (r > n - 2) r = n - 2;
    return r;
}

// The same with the linspace values recomputed (SlTri::affine_points) instead of loaded - no
// dependent memory round trips - in 32-bit indices, the first guess by the reciprocal of the
// spacing (any guess is corrected by the two loops, which then run zero or one step).
SL_HD int64_t sl_rectangle_1d(const SlTri& t, int k, double x) {
    if (!t.affine_points)
        return sl_rectangle_1d(t.points + t.points_off[k], t.grid.num_points[k], t.grid.offset[k],
                               t.grid.unit_maxes[k], x);
    const int n = (int)t.grid.num_points[k];
    const double offset = t.grid.offset[k], unit = t.grid.unit_maxes[k];
    const double g = (x - offset) * t.inv_unit[k];
    int i;
    if (!(g > 0.0)) i = 0; else if (g >= (double)(n - 1)) i = n - 1; else i = (int)g;
    // The guess is at most one point off (|g| < 2^31: its error is far below 1, and the points
    // themselves are i * unit + offset rounded), so ONE step up and ONE step down replace the two
    // search loops - as selects, without divergent control flow (the loops cost ~280 instructions
    // per dimension and lookup in exec-mask bookkeeping).
#define SL_PT(I) ((double)(I) * unit + offset)
    const bool up = (i + 1 < n) && (SL_PT(i + 1) <= x);
    i = up ? i + 1 : i;
    const bool down = (i > 0) && (SL_PT(i) > x);
    i = down ? i - 1 : i;
    int cnt = (SL_PT(i) <= x) ? (i + 1) : 0;      // bins[cnt-1] <= x < bins[cnt]
#undef SL_PT
    if (x != x) cnt = n;                           // NaN sorts last in digitize
    int r = cnt - 1;
    r = r < 0 ? 0 : r;
    r = r > n - 2 ? n - 2 : r;
    return r;
}

SL_HD double sl_fmod_pos(double a, double b) {     // numpy `%` for a >= 0, b > 0
    double r = fmod(a, b);
    return r;
}

// Value (column `col`) of the interpolant at x; also returns the simplex gradient if grad != 0.
SL_HD double sl_tri_eval(const SlTri& t, const double* x, int col, double* grad) {
    const int d = t.grid.d;
    const double eps2 = 2.0 * 2.220446049250313e-16;
    int64_t corner = 0;
    int64_t rk[SL_D];
    double unitc[SL_D], xc[SL_D];
#pragma unroll
    for (int k = 0; k < SL_D; ++k) {
        if (k < d) {
            int64_t r = sl_rectangle_1d(t, k, x[k]);
            rk[k] = r;
            corner += r * t.stride[k];
            // _center_states(clip=True): functions.py:705-712
            double c = x[k] - t.grid.offset[k];
            double lo = 0.0 + eps2, hi = (t.grid.upper[k] - t.grid.offset[k]) - eps2;
            c = (c < lo) ? lo : c;
            c = (c > hi) ? hi : c;
            unitc[k] = sl_fmod_pos(c, t.grid.unit_maxes[k]);       // :1123
            double p = x[k];
            if (t.project) {                                       // :1190-1191 / :1479-1485
                p = (p > t.grid.offset[k]) ? p : t.grid.offset[k];
                p = (p < t.grid.upper[k]) ? p : t.grid.upper[k];
            }
            xc[k] = p;
        }
    }
    // point location inside the unit cell: the simplex whose smallest barycentric weight is largest
    int best = 0;
    double best_min = -1e300;
    for (int s = 0; s < t.nsimplex; ++s) {
        double w0 = 1.0, wmin = 1e300;
        double org[SL_D];
#pragma unroll
        for (int k = 0; k < SL_D; ++k)
            if (k < d) org[k] = ((t.simplices[s][0] >> k) & 1) ? t.grid.unit_maxes[k] : 0.0;
#pragma unroll
        for (int j = 0; j < SL_D; ++j) {
            if (j < d) {
                double w = 0.0;
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < d) w += (unitc[k] - org[k]) * t.hyper[s][k][j];
                w0 -= w;
                wmin = (w < wmin) ? w : wmin;
            }
        }
        wmin = (w0 < wmin) ? w0 : wmin;
        if (wmin > best_min) { best_min = wmin; best = s; }
    }
    // weights relative to the simplex origin in physical coordinates (:1180-1200)
    int64_t v0 = corner;
    double org[SL_D];
    {
        // origin = index_to_state(simplices[:,0]) = (ijk)*unit + offset
#pragma unroll
        for (int k = 0; k < SL_D; ++k) {
            if (k < d) {
                int64_t ik = rk[k];
                int bit = (t.simplices[best][0] >> k) & 1;
                v0 += bit * t.stride[k];
                double tt = (double)(ik + bit) * t.grid.unit_maxes[k];
                org[k] = tt + t.grid.offset[k];
            }
        }
    }
    double w1[SL_D], wsum = 0.0;
#pragma unroll
    for (int j = 0; j < SL_D; ++j) {
        if (j < d) {
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d) w += (xc[k] - org[k]) * t.hyper[best][k][j];
            w1[j] = w;
            wsum += w;
        }
    }
    double p0 = t.table[v0 * t.ncols + col];
    double value = (1.0 - wsum) * p0;
    if (grad) {
#pragma unroll
        for (int k = 0; k < SL_D; ++k) if (k < d) grad[k] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < SL_D; ++j) {
        if (j < d) {
            int64_t vj = corner;
#pragma unroll
            for (int k = 0; k < SL_D; ++k)
                if (k < d) vj += ((t.simplices[best][j + 1] >> k) & 1) * t.stride[k];
            double pj = t.table[vj * t.ncols + col];
            value += w1[j] * pj;
            if (grad) {
#pragma unroll
                for (int k = 0; k < SL_D; ++k)
                    if (k < d) grad[k] += t.hyper[best][k][j] * (pj - p0);
            }
        }
    }
    return value;
}

// exp(x) for x <= 0: 2^k * p(r), k = rint(x log2 e), r = x - k ln 2 in two pieces, p the degree-13
// Taylor polynomial on |r| <= ln 2 / 2 (truncation 4e-18 relative, result within 2 ulp): 20
// instructions and no branches where the library routine takes about twice that.  Used by the
// FP64-VALU-bound Bellman kernels; deep underflow goes through ldexp to 0.
SL_HD double sl_exp_nonpos(double x) {
    x = x < -800.0 ? -800.0 : x;                             // -inf -> 0 (NaN stays NaN)
    const double k = rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;                      // 1/13!
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

// Derived constants of a triangulation; call after filling simplices / hyper / grid.
// ---- candidate simplices per coordinate-order region (host) -----------------------------------
// Both the unit-cell simplices and the d! Kuhn regions {z : z_pi(0) >= z_pi(1) >= ...} are
// simplices with 0/1 vertices.  Simplex S can contain a point of region K in its interior only if
// S and K overlap in a set of positive volume; that is decided exactly enough by enumerating the
// vertices of S n K (all d-subsets of the 2(d+1) facet planes) and testing their affine rank.
namespace sl_tri_detail {
inline bool solve(int n, double a[5][6]) {         // Gauss-Jordan on [A | b], false if singular
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (fabs(a[piv][c]) < 1e-9) return false;
        for (int k = 0; k <= n; ++k) { const double tmp = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = tmp; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r][c] / a[c][c];
            for (int k = c; k <= n; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < n; ++r) a[r][n] /= a[r][r];
    return true;
}
// facet planes g . z + h >= 0 of the simplex with vertices v[0..d] (barycentric coordinates)
inline bool facets(int d, const double v[5][4], double g[5][4], double h[5]) {
    for (int i = 0; i <= d; ++i) {
        // lambda_i(z) = g_i . z + h_i with lambda_i(v_j) = delta_ij
        double a[5][6];
        for (int j = 0; j <= d; ++j) {
            for (int k = 0; k < d; ++k) a[j][k] = v[j][k];
            a[j][d] = 1.0;
            a[j][d + 1] = (i == j) ? 1.0 : 0.0;
        }
        if (!solve(d + 1, a)) return false;
        for (int k = 0; k < d; ++k) g[i][k] = a[k][d + 1];
        h[i] = a[d][d + 1];
    }
    return true;
}
inline bool overlap(int d, const double ga[5][4], const double ha[5], const double gb[5][4],
                    const double hb[5]) {
    const int m = 2 * (d + 1);
    double g[10][4], h[10];
    for (int i = 0; i <= d; ++i) {
        for (int k = 0; k < d; ++k) { g[i][k] = ga[i][k]; g[d + 1 + i][k] = gb[i][k]; }
        h[i] = ha[i]; h[d + 1 + i] = hb[i];
    }
    double verts[256][4];
    int nv = 0;
    int idx[4] = {0, 1, 2, 3};
    // all d-subsets of the m planes
    for (int i = 0; i < d; ++i) idx[i] = i;
    while (true) {
        double a[5][6];
        for (int r = 0; r < d; ++r) {
            for (int k = 0; k < d; ++k) a[r][k] = g[idx[r]][k];
            a[r][d] = -h[idx[r]];
        }
        if (solve(d, a)) {
            bool ok = true;
            for (int q = 0; q < m && ok; ++q) {
                double val = h[q];
                for (int k = 0; k < d; ++k) val += g[q][k] * a[k][d];
                ok = val >= -1e-9;
            }
            if (ok && nv < 256) { for (int k = 0; k < d; ++k) verts[nv][k] = a[k][d]; ++nv; }
        }
        int r = d - 1;
        while (r >= 0 && idx[r] == m - d + r) --r;
        if (r < 0) break;
        ++idx[r];
        for (int q = r + 1; q < d; ++q) idx[q] = idx[q - 1] + 1;
    }
    if (nv < d + 1) return false;
    // affine rank of the feasible vertices
    double b[256][4];
    for (int i = 1; i < nv; ++i) for (int k = 0; k < d; ++k) b[i - 1][k] = verts[i][k] - verts[0][k];
    int rank = 0, rows = nv - 1;
    for (int c = 0; c < d && rank < rows; ++c) {
        int piv = rank;
        for (int r = rank + 1; r < rows; ++r) if (fabs(b[r][c]) > fabs(b[piv][c])) piv = r;
        if (fabs(b[piv][c]) < 1e-7) continue;
        for (int k = 0; k < d; ++k) { const double tmp = b[rank][k]; b[rank][k] = b[piv][k]; b[piv][k] = tmp; }
        for (int r = rank + 1; r < rows; ++r) {
            const double f = b[r][c] / b[rank][c];
            for (int k = c; k < d; ++k) b[r][k] -= f * b[rank][k];
        }
        ++rank;
    }
    return rank == d;
}
}  // namespace sl_tri_detail

// bit per pair (i < j), pairs in lexicographic order: 1 iff z_i < z_j
SL_HD int sl_tri_region_code(int d, const double* z) {
    int code = 0, bit = 0;
    for (int i = 0; i < d; ++i)
        for (int j = i + 1; j < d; ++j, ++bit) code |= (z[i] < z[j]) ? (1 << bit) : 0;
    return code;
}

// bit per pair (i < j), same order as sl_tri_region_code: 1 iff z_i + z_j < 1
SL_HD int sl_tri_sum_code(int d, const double* z) {
    int code = 0, bit = 0;
    for (int i = 0; i < d; ++i)
        for (int j = i + 1; j < d; ++j, ++bit) code |= (z[i] + z[j] < 1.0) ? (1 << bit) : 0;
    return code;
}

// Smallest-weight margin above which the candidate shortcut of sl_tri_value_fast is accepted.
// 0: every point that is strictly interior to a candidate simplex in computed arithmetic.  For a
// point within rounding of a shared face the computed minimum of a simplex OUTSIDE the candidate
// list may be larger by a few ulps, so there the shortcut and the full arg-max walk can pick
// different simplices that both contain the point up to rounding: the interpolated values then
// agree to rounding, not bit for bit (the reference's own choice on faces depends on SciPy's
// search history; the parity tests exclude such points).  A margin of 1e-12 sends all of them to
// the full walk - 20 % of the lookups of the 64^4 cart-pole sweep, whose out-of-range successors
// are projected onto the boundary faces: 27.8 -> 33.3 ms per sweep (profiles/r03_summary.md).
#ifndef SL_TRI_SHORTCUT_MARGIN
#define SL_TRI_SHORTCUT_MARGIN 0.0
#endif
inline void sl_tri_regions_compute(SlTri& t);
// (host time: the exact first-level table is milliseconds; the sampled second level of a 4-D table
// walks 2e6 points x up to 24 simplices, 0.5-1 s on the first 4-D sl_tri_set of a thread.  The
// result is kept per dimension and reused while the unit-cell simplices stay the same, e.g. across
// the uploads of a value-iteration loop)
inline void sl_tri_regions(SlTri& t) {
    // one remembered result per dimension: programs that alternate between, say, a 4-D value
    // table and a 2-D one do not pay the 4-D sampling again
    static thread_local SlTri memos[SL_D + 1];
    static thread_local bool have[SL_D + 1] = {false};
    const int d = t.grid.d;
    if (d < 2 || d > 4) {                            // nothing to prune
        for (int c = 0; c < SL_TRI_CODES; ++c) t.ncand[c] = 0;
        t.has_fine = 0;
        return;
    }
    SlTri& memo = memos[d];
    if (have[d] && memo.nsimplex == t.nsimplex &&
        memcmp(memo.simplices, t.simplices, sizeof(t.simplices)) == 0) {
        memcpy(t.ncand, memo.ncand, sizeof(t.ncand));
        memcpy(t.cand, memo.cand, sizeof(t.cand));
        t.has_fine = memo.has_fine;
        memcpy(t.perm_index, memo.perm_index, sizeof(t.perm_index));
        memcpy(t.fine, memo.fine, sizeof(t.fine));
        return;
    }
    sl_tri_regions_compute(t);
    memo.nsimplex = t.nsimplex;
    memcpy(memo.simplices, t.simplices, sizeof(t.simplices));
    memcpy(memo.ncand, t.ncand, sizeof(t.ncand));
    memcpy(memo.cand, t.cand, sizeof(t.cand));
    memo.has_fine = t.has_fine;
    memcpy(memo.perm_index, t.perm_index, sizeof(t.perm_index));
    memcpy(memo.fine, t.fine, sizeof(t.fine));
    have[d] = true;
}

inline void sl_tri_regions_compute(SlTri& t) {
    using namespace sl_tri_detail;
    const int d = t.grid.d;
    for (int c = 0; c < SL_TRI_CODES; ++c) t.ncand[c] = 0;
    t.has_fine = 0;
    if (d < 2 || d > 4 || t.nsimplex < 3) return;           // nothing to prune
    double sg[SL_MAX_SIMPLICES][5][4], sh[SL_MAX_SIMPLICES][5];
    for (int s = 0; s < t.nsimplex; ++s) {
        double v[5][4];
        for (int q = 0; q <= d; ++q)
            for (int k = 0; k < d; ++k) v[q][k] = (double)((t.simplices[s][q] >> k) & 1);
        if (!facets(d, v, sg[s], sh[s])) return;             // degenerate simplex: keep the full walk
    }
    int perm[4] = {0, 1, 2, 3};
    bool more = true;
    uint8_t ncand[SL_TRI_CODES];
    for (int c = 0; c < SL_TRI_CODES; ++c) ncand[c] = 0;
    while (more) {
        // Kuhn region z_perm[0] >= z_perm[1] >= ... : vertices 0, e_p0, e_p0 + e_p1, ...
        double v[5][4], kg[5][4], kh[5], centre[4];
        for (int k = 0; k < d; ++k) v[0][k] = 0.0;
        for (int q = 1; q <= d; ++q) {
            for (int k = 0; k < d; ++k) v[q][k] = v[q - 1][k];
            v[q][perm[q - 1]] = 1.0;
        }
        for (int k = 0; k < d; ++k) {
            centre[k] = 0.0;
            for (int q = 0; q <= d; ++q) centre[k] += v[q][k] / (d + 1);
        }
        const int code = sl_tri_region_code(d, centre);
        if (!facets(d, v, kg, kh)) return;
        for (int s = 0; s < t.nsimplex; ++s) {
            if (!overlap(d, sg[s], sh[s], kg, kh)) continue;
            if (ncand[code] >= SL_TRI_MAXCAND) return;         // too many: keep the full walk
            t.cand[code][ncand[code]++] = (uint8_t)s;
        }
        if (ncand[code] == 0) return;
        // next permutation of perm[0..d)
        int i = d - 2;
        while (i >= 0 && perm[i] > perm[i + 1]) --i;
        if (i < 0) more = false;
        else {
            int j = d - 1;
            while (perm[j] < perm[i]) --j;
            int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
            for (int a = i + 1, b = d - 1; a < b; ++a, --b) { tmp = perm[a]; perm[a] = perm[b]; perm[b] = tmp; }
        }
    }
    for (int c = 0; c < SL_TRI_CODES; ++c) t.ncand[c] = ncand[c];
    // ---- second level by sampling (4-D) ------------------------------------------------------
    if (d != 4) return;
    int nperm = 0;
    for (int c = 0; c < SL_TRI_CODES; ++c) t.perm_index[c] = (ncand[c] > 0) ? (uint8_t)nperm++ : 0xFF;
    if (nperm != 24) return;
    memset(t.fine, 0xFF, sizeof(t.fine));
    // hits[region][simplex]: how many of the sampled points of the region lie in the simplex; the
    // four most frequent simplices of a region become its candidates (in ascending index order:
    // the first-index tie rule), rarer slivers are left to the full walk
    static const int NSAMPLES = 2000000;
    static thread_local uint32_t hits[24 * 64][SL_MAX_SIMPLICES];
    memset(hits, 0, sizeof(hits));
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (int n = 0; n < NSAMPLES; ++n) {
        double z[4];
        for (int k = 0; k < 4; ++k) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            z[k] = ((double)(rng >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        }
        int best = -1;
        double best_min = 1e-9;                       // strictly interior points only
        for (int s = 0; s < t.nsimplex; ++s) {
            double wmin = 1e300;
            for (int i = 0; i <= 4 && wmin > best_min; ++i) {
                double w = sh[s][i];
                for (int k = 0; k < 4; ++k) w += sg[s][i][k] * z[k];
                wmin = w < wmin ? w : wmin;
            }
            if (wmin > best_min) { best_min = wmin; best = s; break; }   // interiors are disjoint
        }
        if (best < 0) continue;
        ++hits[t.perm_index[sl_tri_region_code(4, z)] * 64 + sl_tri_sum_code(4, z)][best];
    }
    for (int idx = 0; idx < 24 * 64; ++idx) {
        int chosen[4] = {-1, -1, -1, -1};
        for (int q = 0; q < 4; ++q) {
            int arg = -1;
            uint32_t top = 0;
            for (int sidx = 0; sidx < t.nsimplex; ++sidx) {
                bool taken = false;
                for (int p = 0; p < q; ++p) taken = taken || chosen[p] == sidx;
                if (!taken && hits[idx][sidx] > top) { top = hits[idx][sidx]; arg = sidx; }
            }
            chosen[q] = arg;
        }
        for (int a = 0; a < 4; ++a)                          // ascending, -1 (none) last
            for (int b2 = a + 1; b2 < 4; ++b2)
                if (chosen[b2] >= 0 && (chosen[a] < 0 || chosen[b2] < chosen[a])) {
                    const int tmp = chosen[a]; chosen[a] = chosen[b2]; chosen[b2] = tmp;
                }
        for (int q = 0; q < 4; ++q) t.fine[idx][q] = chosen[q] < 0 ? 0xFF : (uint8_t)chosen[q];
    }
    t.has_fine = 1;
}

inline void sl_tri_finish(SlTri& t, const double* h_points) {
    const int d = t.grid.d;
    // are the discrete points np.linspace's `i * step + start` (functions.py:565-567)?
    t.affine_points = 1;
    for (int k = 0; k < d; ++k) {
        t.inv_unit[k] = 1.0 / t.grid.unit_maxes[k];
        if (t.grid.num_points[k] >= (1ll << 31)) t.affine_points = 0;
    }
    for (int k = 0; k < d && t.affine_points; ++k)
        for (int64_t i = 0; i < t.grid.num_points[k]; ++i) {
            const double pt = (double)i * t.grid.unit_maxes[k] + t.grid.offset[k];
            if (!(pt == h_points[t.points_off[k] + i])) { t.affine_points = 0; break; }
        }
    for (int s = 0; s < t.nsimplex; ++s)
        for (int j = 0; j < d; ++j) {
            double c = 0.0;
            for (int k = 0; k < d; ++k)
                if ((t.simplices[s][0] >> k) & 1) c = fma(t.grid.unit_maxes[k], t.hyper[s][k][j], c);
            t.hyper_c[s][j] = c;
        }
    sl_tri_regions(t);
}

// a mod b for a >= 0, b > 0, exactly as fmod: the remainder is representable, so one fused
// multiply-add with the right integer quotient returns it without rounding.
SL_HD double sl_fmod_exact(double a, double b) {
    double q = floor(a / b);
    double r = fma(-q, b, a);
    if (r < 0.0) { q -= 1.0; r = fma(-q, b, a); }
    else if (r >= b) { q += 1.0; r = fma(-q, b, a); }
    return r;
}
// the quotient guessed with a reciprocal (inv_b ~ 1 / b) and corrected until 0 <= r < b
SL_HD double sl_fmod_exact(double a, double b, double inv_b) {
    // |q| < 2^31 here: the reciprocal's guess is at most one off on either side
    double q = floor(a * inv_b);
    double r = fma(-q, b, a);
    const double qm = q - 1.0, rm = fma(-qm, b, a);
    q = (r < 0.0) ? qm : q;
    r = (r < 0.0) ? rm : r;
    const double qp = q + 1.0, rp = fma(-qp, b, a);
    r = (r >= b) ? rp : r;
    return r;
}

// Column 0 of the interpolant at x for a compile-time dimension (the Bellman sweeps evaluate it
// A times per vertex).  Same rule as sl_tri_eval - the unit-cell simplex whose smallest
// barycentric weight is largest - with the weights as fused w = G_s u - c_s; where several
// simplices contain the point (a shared face) the candidates agree on the value.
// located point: the D + 1 table rows of its simplex and their weights (the origin's weight is
// 1 - sum of the others, w[0] here)
template <int D>
struct SlTriLoc {
    int64_t row[D + 1];          // row[0] = simplex origin
    double  w[D + 1];            // w[0] = 1 - (w[1] + ... + w[D])
    int64_t corner;              // vertex index of the rectangle's lower corner and the unit-cell
    int     simplex;             // simplex: (corner, simplex, w[1..D]) rebuild the rest (sl_tri_reloc)
};

template <int DT>
SL_HD void sl_tri_locate_fast(const SlTri& t, const double* x, SlTriLoc<(DT > 0 ? DT : 1)>& loc) {
    constexpr int D = DT > 0 ? DT : 1;
    const double eps2 = 2.0 * 2.220446049250313e-16;
    int64_t corner = 0;
    double base[D], unitc[D], xc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int64_t r = sl_rectangle_1d(t, k, x[k]);
        corner += r * t.stride[k];
        base[k] = (double)r;
        double c = x[k] - t.grid.offset[k];
        const double lo = 0.0 + eps2, hi = (t.grid.upper[k] - t.grid.offset[k]) - eps2;
        c = (c < lo) ? lo : c;
        c = (c > hi) ? hi : c;
        unitc[k] = sl_fmod_exact(c, t.grid.unit_maxes[k], t.inv_unit[k]);
        double pp = x[k];
        if (t.project) {
            pp = (pp > t.grid.offset[k]) ? pp : t.grid.offset[k];
            pp = (pp < t.grid.upper[k]) ? pp : t.grid.upper[k];
        }
        xc[k] = pp;
    }
    int best = 0;
    double best_min = -1e300;
#define SL_TRI_TRY(S)                                                                           \
    do {                                                                                        \
        const int s_ = (S);                                                                     \
        double w0 = 1.0, wmin = 1e300;                                                          \
        _Pragma("unroll") for (int j = 0; j < D; ++j) {                                         \
            double w = -t.hyper_c[s_][j];                                                       \
            _Pragma("unroll") for (int k = 0; k < D; ++k) w = fma(unitc[k], t.hyper[s_][k][j], w); \
            w0 -= w;                                                                            \
            wmin = fmin(wmin, w);                                                               \
        }                                                                                       \
        wmin = fmin(wmin, w0);                                                                  \
        if (wmin > best_min) { best_min = wmin; best = s_; }                                    \
    } while (0)
    // The simplices overlapping the point's coordinate-order region first: a strictly positive
    // smallest weight means the point is interior to that simplex, which then is the maximiser
    // over all simplices too.  Anything else (points on faces, exact ties: 0.6 % of the lookups of
    // the 64^4 Bellman sweep, 2.2 % of its wavefront steps) takes the full walk.
    bool full = true;
    if (D >= 2 && D <= 4) {
        double z[D];
#pragma unroll
        for (int k = 0; k < D; ++k) z[k] = unitc[k] * t.inv_unit[k];
        const int code = sl_tri_region_code(D, z);
        // second level (4-D): at most four candidates; two per round - their hyperplane rows are
        // read together instead of in two dependent round trips (an odd last candidate is tried
        // twice, which changes nothing)
        uint32_t fw = 0xFFFFFFFFu;
        if (D == 4 && t.has_fine)
            fw = *reinterpret_cast<const uint32_t*>(t.fine[t.perm_index[code] * 64 + sl_tri_sum_code(D, z)]);
        int nc = 1;
        if ((fw & 0xFFu) != 0xFFu) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int s0 = (int)((fw >> (16 * r)) & 0xFFu);
                if (s0 == 0xFF) break;
                int s1 = (int)((fw >> (16 * r + 8)) & 0xFFu);
                s1 = (s1 == 0xFF) ? s0 : s1;
                SL_TRI_TRY(s0);
                SL_TRI_TRY(s1);
            }
        } else {
            nc = t.ncand[code];
            for (int i = 0; i < nc; i += 2) {
                const int s0 = t.cand[code][i], s1 = t.cand[code][i + 1 < nc ? i + 1 : i];
                SL_TRI_TRY(s0);
                SL_TRI_TRY(s1);
            }
        }
        full = !(nc > 0 && best_min > SL_TRI_SHORTCUT_MARGIN);
    }
    if (full) {
        best = 0;
        best_min = -1e300;
#pragma unroll 2
        for (int s = 0; s < t.nsimplex; ++s) SL_TRI_TRY(s);
    }
#undef SL_TRI_TRY
    // weights relative to the simplex origin in physical coordinates (functions.py:1180-1200)
    const int code0 = t.simplices[best][0];
    int64_t v0 = corner;
    double rel[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int bit = (code0 >> k) & 1;
        v0 += bit * t.stride[k];
        const double tt = (base[k] + (double)bit) * t.grid.unit_maxes[k];
        rel[k] = xc[k] - (tt + t.grid.offset[k]);
    }
    double wsum = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) w = fma(rel[k], t.hyper[best][k][j], w);
        const int code = t.simplices[best][j + 1];
        int64_t vj = corner;
#pragma unroll
        for (int k = 0; k < D; ++k) vj += ((code >> k) & 1) * t.stride[k];
        loc.row[j + 1] = vj * t.ncols;
        loc.w[j + 1] = w;
        wsum += w;
    }
    loc.row[0] = v0 * t.ncols;
    loc.w[0] = 1.0 - wsum;
    loc.corner = corner;
    loc.simplex = best;
}

// A located point from what the successor cache keeps of it (sl_succ.hip): the same rows and the
// same weights, bit for bit, as sl_tri_locate_fast produced - w[0] by the same ordered sum.
template <int D>
SL_HD void sl_tri_reloc(const SlTri& t, int64_t corner, int simplex, const double* w, SlTriLoc<D>& loc) {
    double wsum = 0.0;
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        const int code = t.simplices[simplex][j];
        int64_t v = corner;
#pragma unroll
        for (int k = 0; k < D; ++k) v += ((code >> k) & 1) * t.stride[k];
        loc.row[j] = v * t.ncols;
        if (j > 0) {
            loc.w[j] = w[j - 1];
            wsum += w[j - 1];
        }
    }
    loc.w[0] = 1.0 - wsum;
    loc.corner = corner;
    loc.simplex = simplex;
}

// the table reads of a located point, apart from their use: kernels issue them early
template <int D>
SL_HD void sl_tri_gather(const SlTri& t, const SlTriLoc<D>& loc, double* vals) {
#pragma unroll
    for (int j = 0; j <= D; ++j) vals[j] = t.table[loc.row[j]];
}

template <int D>
SL_HD double sl_tri_combine(const SlTriLoc<D>& loc, const double* vals) {
    double acc = 0.0;
#pragma unroll
    for (int j = 1; j <= D; ++j) acc = fma(loc.w[j], vals[j], acc);
    return fma(loc.w[0], vals[0], acc);
}

template <int DT>
SL_HD double sl_tri_value_fast(const SlTri& t, const double* x) {
    if (DT == 0) return sl_tri_eval(t, x, 0, nullptr);
    constexpr int D = DT > 0 ? DT : 1;
    SlTriLoc<D> loc;
    double vals[D + 1];
    sl_tri_locate_fast<DT>(t, x, loc);
    sl_tri_gather<D>(t, loc, vals);
    return sl_tri_combine<D>(loc, vals);
}

// ---------------------------------------------------------------------------------------------
// LyapunovNetwork forward (+ input gradient)   examples/utilities.py:85-104
// ---------------------------------------------------------------------------------------------
#define SL_NN_MAXW 64
// tanh of the network activations.  On the device: (1 - e) / (1 + e) with e = exp(-2 |x|) from the
// polynomial of sl_exp_nonpos, written as -em1 / (2 + em1) with em1 = e - 1 taken from the polynomial
// itself where the argument needs no scaling (|x| < 0.17: no cancellation) - within 4 ulp of the
// library routine at a quarter of its 165 instructions (a LyapunovNetwork sweep spends more vector
// instructions on its 96 activations per lane and tile than on anything else).  The host build (the
// bit-exactness simulator of tests/hostsim) keeps libm's.
SL_HD double sl_tanh(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = -2.0 * fabs(x);
    y = y < -80.0 ? -80.0 : y;                               // tanh = 1 beyond |x| = 40 (NaN stays NaN)
    const double k = rint(y * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, y);
    r = fma(k, -1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;                      // 1/13!
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
    q = fma(q, r, 1.0);                                      // (exp(r) - 1) / r
    const double e = ldexp(fma(q, r, 1.0), (int)k);
    const double em1 = (k == 0.0) ? q * r : e - 1.0;
    const double t = (0.0 - em1) / (2.0 + em1);
    return copysign(t, x);
#else
    return tanh(x);
#endif
}
SL_HD double sl_act(int a, double x) { return a == 1 ? sl_tanh(x) : (a == 2 ? (x > 0.0 ? x : 0.0) : x); }
SL_HD double sl_dact(int a, double pre, double post) {
    return a == 1 ? (1.0 - post * post) : (a == 2 ? (pre > 0.0 ? 1.0 : 0.0) : 1.0);
}

// ---------------------------------------------------------------------------------------------
// the model as the kernels see it
// ---------------------------------------------------------------------------------------------
struct SlGpHeadDev {
    int32_t n, n_pad, p, dout, col0, nslab2, reserved0, reserved1;
    double  variance;
    double  inv_ls[SL_P];
    const double* xs;        // [p][n_pad], inputs already divided by the lengthscales
    const double* mpack;     // MFMA A-fragments of Linv, see sl_gp.hip
    const double* alpha;     // [n_pad][dout]  alpha' = Linv^T alpha
    // sum-of-products kernel (sl_gp_set_head_kernel): device copy of the description, or null for
    // the RBF head of sl_gp_set_head.  xs then holds the inputs unscaled and inv_ls is all ones.
    const sl_gp_kernel* kernel;
};

// k(a, b) of a sum-of-products kernel (gpflow 0.4.0 kernels.py: Add.K / Prod.K over RBF.K,
// Matern32.K, Linear.K; the description of include/sl_hip.h).  a, b: unscaled inputs.
SL_HD double sl_kernel_eval(const sl_gp_kernel& ks, int p, const double* a, const double* b) {
    double total = 0.0, prod = 1.0;
    int cur = 0;
    for (int f = 0; f < ks.nfactors; ++f) {
        const sl_gp_kernel_factor& fac = ks.factor[f];
        if (fac.product != cur) {
            total += prod;
            prod = 1.0;
            cur = fac.product;
        }
        double v = 0.0;
        if (fac.kind == SL_KERNEL_LINEAR) {
#pragma unroll
            for (int q = 0; q < SL_P; ++q)
                if (q < p) v = fma(a[q] * fac.variance[q], b[q], v);
        } else {
            double r2 = 0.0;
#pragma unroll
            for (int q = 0; q < SL_P; ++q) {
                if (q < p) {
                    const double dq = (a[q] - b[q]) * fac.inv_lengthscales[q];
                    r2 = fma(dq, dq, r2);
                }
            }
            if (fac.kind == SL_KERNEL_RBF) {
                v = fac.variance[0] * sl_exp_nonpos(-0.5 * r2);
            } else {                                    // Matern32, euclid_dist's 1e-12
                const double r = 1.7320508075688772 * sqrt(r2 + 1e-12);
                v = fac.variance[0] * (1.0 + r) * sl_exp_nonpos(-r);
            }
        }
        prod *= v;
    }
    return total + prod;
}
// a wavefront-uniform int as a scalar (the factor kinds of a kernel description read from LDS:
// the branches on them stay scalar branches)
#if defined(__HIP_DEVICE_COMPILE__)
#define SL_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#else
#define SL_UNIFORM_INT(x) (x)
#endif

// k(a0, b) and k(a1, b) in one walk over the factors (k_gp_small: the two training points a lane
// holds of a slab pair; the description is read once for both - from the workgroup's LDS copy).
// The same operations in the same order as sl_kernel_eval for each of the two.
SL_HD void sl_kernel_eval2(const sl_gp_kernel& ks, int p, const double* a0, const double* a1,
                           const double* b, double* k0, double* k1) {
    double total0 = 0.0, prod0 = 1.0, total1 = 0.0, prod1 = 1.0;
    int cur = 0;
    const int nfactors = SL_UNIFORM_INT(ks.nfactors);
    for (int f = 0; f < nfactors; ++f) {
        const sl_gp_kernel_factor& fac = ks.factor[f];
        const int kind = SL_UNIFORM_INT(fac.kind), product = SL_UNIFORM_INT(fac.product);
        if (product != cur) {
            total0 += prod0; total1 += prod1;
            prod0 = 1.0; prod1 = 1.0;
            cur = product;
        }
        double v0 = 0.0, v1 = 0.0;
        if (kind == SL_KERNEL_LINEAR) {
#pragma unroll
            for (int q = 0; q < SL_P; ++q) {
                if (q < p) {
                    const double var = fac.variance[q];
                    v0 = fma(a0[q] * var, b[q], v0);
                    v1 = fma(a1[q] * var, b[q], v1);
                }
            }
        } else {
            double r0 = 0.0, r1 = 0.0;
#pragma unroll
            for (int q = 0; q < SL_P; ++q) {
                if (q < p) {
                    const double inv = fac.inv_lengthscales[q];
                    const double d0 = (a0[q] - b[q]) * inv, d1 = (a1[q] - b[q]) * inv;
                    r0 = fma(d0, d0, r0);
                    r1 = fma(d1, d1, r1);
                }
            }
            const double var = fac.variance[0];
            if (kind == SL_KERNEL_RBF) {
                v0 = var * sl_exp_nonpos(-0.5 * r0);
                v1 = var * sl_exp_nonpos(-0.5 * r1);
            } else {                                    // Matern32, euclid_dist's 1e-12
                const double s0 = 1.7320508075688772 * sqrt(r0 + 1e-12);
                const double s1 = 1.7320508075688772 * sqrt(r1 + 1e-12);
                v0 = var * (1.0 + s0) * sl_exp_nonpos(-s0);
                v1 = var * (1.0 + s1) * sl_exp_nonpos(-s1);
            }
        }
        prod0 *= v0;
        prod1 *= v1;
    }
    *k0 = total0 + prod0;
    *k1 = total1 + prod1;
}
// k(x, x) as kern.Kdiag states it (Stationary.Kdiag: the variance itself, not K through
// euclid_dist; Linear.Kdiag: sum_q variance_q x_q^2)
SL_HD double sl_kernel_diag(const sl_gp_kernel& ks, int p, const double* x) {
    double total = 0.0, prod = 1.0;
    int cur = 0;
    for (int f = 0; f < ks.nfactors; ++f) {
        const sl_gp_kernel_factor& fac = ks.factor[f];
        if (fac.product != cur) {
            total += prod;
            prod = 1.0;
            cur = fac.product;
        }
        double v = fac.variance[0];
        if (fac.kind == SL_KERNEL_LINEAR) {
            v = 0.0;
#pragma unroll
            for (int q = 0; q < SL_P; ++q)
                if (q < p) v = fma(x[q] * x[q], fac.variance[q], v);
        }
        prod *= v;
    }
    return total + prod;
}

struct SlGpDev {
    int32_t nheads, reserved;
    double  beta;
    SlGpHeadDev head[SL_MAX_GP_HEADS];
};

struct SlDevModel {
    sl_model_desc m;
    SlGridFast    gf;
    int32_t       in_dim;      // d + m
    int32_t       uncertain;   // dynamics returns (mean, error)
};

// Dimensions of one kernel instantiation.  Kernels are compiled for fixed (state, action)
// dimensions (DT, MT > 0) so that every `k < d` below folds at compile time, plus one generic
// variant (DT = MT = 0) that reads them from the model.
struct SlDims { int d, m, p; };
template <int DT, int MT>
SL_HD SlDims sl_dims(const SlDevModel& M) {
    SlDims r;
    r.d = DT > 0 ? DT : M.m.grid.d;
    r.m = MT > 0 ? MT : M.m.policy.m;
    r.p = r.d + r.m;
    return r;
}

// L_v(z): writes lv_cols values
SL_HD void sl_lv(const SlDevModel& M, int d, const double* z, double* lv) {
    const sl_lipschitz_desc& l = M.m.lipschitz;
    if (l.lv_kind == SL_LIP_CONST) { lv[0] = l.lv_const; return; }
    double t[SL_D];
    sl_rows_dot<SL_D, SL_D>(l.lv_matrix, d, d, z, t);
    if (l.lv_kind == SL_LIP_ABS_LINEAR) {
#pragma unroll
        for (int j = 0; j < SL_D; ++j) if (j < d) lv[j] = fabs(t[j]);
    } else {   // SL_LIP_NORM_LINEAR
        double acc = fabs(t[0]);
#pragma unroll
        for (int j = 1; j < SL_D; ++j) if (j < d) acc = acc + fabs(t[j]);
        lv[0] = acc;
    }
}

// lyapunov.py:282-288
// -|L_v(x)|_1 (1 + L_f(x)) tau, lyapunov.py:265-288; L_f a scalar or c + ||M x||_1 (:227-244)
SL_HD double sl_threshold(const SlDevModel& M, int d, const double* lv_x, double tau,
                          const double* x) {
    const sl_lipschitz_desc& l = M.m.lipschitz;
    double lf = l.lf_const;
    if (l.lf_kind == SL_LF_AFFINE_NORM1) {
        double t[SL_D];
        sl_rows_dot<SL_D, SL_D>(l.lf_matrix, d, d, x, t);
        double acc = fabs(t[0]);
#pragma unroll
        for (int j = 1; j < SL_D; ++j) if (j < d) acc = acc + fabs(t[j]);
        lf = lf + acc;
    }
    double l1 = lv_x[0];
    if ((l.lv_kind == SL_LIP_ABS_LINEAR || l.lv_kind == SL_LIP_ABS_GRAD) && d > 1) {
        l1 = fabs(lv_x[0]);
#pragma unroll
        for (int j = 1; j < SL_D; ++j) if (j < d) l1 = l1 + fabs(lv_x[j]);
    }
    double t = (-l1) * (1.0 + lf);
    return t * tau;
}

// lyapunov.py:344-352, 376: (V(next) - V(x)) + sum_j lv_j(next) err_j
SL_HD double sl_decrease(const SlDevModel& M, int d, double v_x, double v_next,
                         const double* lv_next, const double* err) {
    double dv = v_next - v_x;
    double bound = 0.0;
    if (M.uncertain) {
        const sl_lipschitz_desc& l = M.m.lipschitz;
        const bool bcast = !(l.lv_kind == SL_LIP_ABS_LINEAR || l.lv_kind == SL_LIP_ABS_GRAD) || d == 1;
        bound = lv_next[0] * err[0];
#pragma unroll
        for (int j = 1; j < SL_D; ++j) {
            if (j < d) {
                double t = (bcast ? lv_next[0] : lv_next[j]) * err[j];
                bound = bound + t;
            }
        }
    }
    return dv + bound;
}

// policy(x) for the closed-form kinds (TABLE / TRI handled by the caller)
SL_HD void sl_policy_closed_form(const SlDevModel& M, SlDims n, const double* x, double* u) {
    const sl_policy_desc& p = M.m.policy;
    if (p.kind == SL_POLICY_LINEAR) {
        sl_rows_dot<SL_M, SL_D>(p.matrix, n.m, n.d, x, u);
    } else {
#pragma unroll
        for (int a = 0; a < SL_M; ++a) if (a < n.m) u[a] = p.constant[a];
    }
    sl_saturate(p, n.m, u);
}

// z[d + a] = u[a] without run-time register indexing
SL_HD void sl_append_action(SlDims n, const double* u, double* z) {
#pragma unroll
    for (int q = 0; q < SL_P; ++q) {
#pragma unroll
        for (int a = 0; a < SL_M; ++a)
            if (a < n.m && q == n.d + a) z[q] = u[a];
    }
}

// deterministic dynamics f(z), z = [x, u] (an SL_P-sized array).  DYN > 0 fixes the kind at
// compile time (the kernels are instantiated per kind so that the other kinds' constants and code
// do not occupy scalar registers).
template <int DYN = 0>
SL_HD void sl_dynamics_det(const SlDevModel& M, SlDims n, const double* z, double* nxt) {
    const sl_dynamics_desc& f = M.m.dynamics;
    const int kind = DYN > 0 ? DYN : f.kind;
    if (kind == SL_DYN_PENDULUM || kind == SL_DYN_CARTPOLE) {
        double u0 = 0.0;
#pragma unroll
        for (int q = 0; q < SL_P; ++q) if (q == n.d) u0 = z[q];
        if (kind == SL_DYN_PENDULUM) sl_pendulum(f, z, &u0, nxt); else sl_cartpole(f, z, &u0, nxt);
        return;
    }
    sl_rows_dot<SL_D, SL_P>(f.matrix, n.d, n.p, z, nxt);
}

// (V, index) key: ascending V, -0 == +0, NaN last (numpy sort order), ties by index
SL_HD uint64_t sl_vbits(double v) {
    if (v != v) return 0xffffffffffffffffull;
    if (v == 0.0) v = 0.0;
    union { double d; uint64_t u; } c;
    c.d = v;
    return (c.u & 0x8000000000000000ull) ? ~c.u : (c.u | 0x8000000000000000ull);
}
SL_HD double sl_vbits_to_double(uint64_t b) {
    union { double d; uint64_t u; } c;
    if (b == 0xffffffffffffffffull) { c.u = 0x7ff8000000000000ull; return c.d; }
    c.u = (b & 0x8000000000000000ull) ? (b & 0x7fffffffffffffffull) : ~b;
    return c.d;
}
SL_HD bool sl_key_less(uint64_t va, int64_t ia, uint64_t vb, int64_t ib) {
    return (va < vb) || (va == vb && ia < ib);
}

// =============================================================================================
// V of 8 consecutive cells of a grid row, recomputed from the flat index (sl_level.hip)
// =============================================================================================
#define SL_ROW_CELLS 8

// May the ordering keys V(all_points[i]) (lyapunov.py:305-322, 512) be recomputed from the cell
// index instead of read?  Quadratic V, 1..4 dimensions, a last axis of whole bytes of the bit masks
// (a thread's 8 cells share a row), and np.linspace points (last point = the upper limit,
// functions.py:612-638) that equal index_to_state (functions.py:728-731) bit for bit.
SL_HD bool sl_values_implicit_ok(const sl_model_desc& m) {
    const int d = m.grid.d;
    if (m.value.kind != SL_V_QUADRATIC || d < 1 || d > 4) return false;
    if (m.grid.num_points[d - 1] % SL_ROW_CELLS) return false;
    for (int k = 0; k < d; ++k) {
        volatile double t = (double)(m.grid.num_points[k] - 1) * m.grid.unit_maxes[k];
        volatile double s = t + m.grid.offset[k];
        if (s != m.grid.upper[k]) return false;
    }
    return true;
}

// sl_vbits (order-preserving float64 -> uint64, -0 = +0, NaN last) in eight instructions
SL_HD uint64_t sl_vbits_fast(double v) {
    const double z = v + 0.0;                             // -0 -> +0 (round to nearest); NaN stays
    union { double d; uint64_t u; } c;
    c.d = z;
    const uint64_t flip = (uint64_t)((int64_t)c.u >> 63) | 0x8000000000000000ull;
    return (z != z) ? ~0ull : (c.u ^ flip);
}

// DT = 0: read from `values`; DT = 1..4: quadratic V of a DT-dimensional grid from the index - the
// ordered sums of functions.py:1534-1539 with the prefix over the leading coordinates shared by the
// row (the same numbers, rounding included, as sl_quadratic on sl_index_to_grid_point)
template <int DT>
struct SlRowValues {
    static constexpr int D = DT > 0 ? DT : 1, L = D - 1;
    double lin_pre[D], x[D];
    int64_t ijk[SL_D];

    SL_HD void point(const SlDevModel& M, int k) {
        const double t = (double)(int)ijk[k] * M.m.grid.unit_maxes[k];               // functions.py:731
        const double s = t + M.m.grid.offset[k];
        x[k] = (ijk[k] == M.m.grid.num_points[k] - 1) ? M.m.grid.upper[k] : s;       // np.linspace
    }

    SL_HD void start_row(const SlDevModel& M, int64_t idx) {
        sl_unravel(M.m.grid, M.gf, D, idx, ijk);
#pragma unroll
        for (int k = 0; k < L; ++k) point(M, k);
        if (L > 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double b = x[0] * M.m.value.matrix[0][j];
#pragma unroll
                for (int k = 1; k < L; ++k) { const double t = x[k] * M.m.value.matrix[k][j]; b = b + t; }
                lin_pre[j] = b;
            }
        }
    }

    SL_HD double cell(const SlDevModel& M) {
        point(M, L);
        double vx = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double tc = x[L] * M.m.value.matrix[L][j];
            const double lin = L > 0 ? lin_pre[j] + tc : tc;
            const double q = lin * x[j];
            vx = (j == 0) ? q : (vx + q);
        }
        return M.m.value.negate ? (vx * -1.0) : vx;
    }

    // V of cells i0 .. i0 + 7.  DT > 0: the eight cells lie in ONE row of the last axis
    // (sl_values_implicit_ok), so the row is unravelled once and there is no carry.
    SL_HD void eight(const SlDevModel& M, const double* values, int64_t lo, int64_t hi, int64_t i0,
                     double* v8) {
        if (DT == 0) {
#pragma unroll
            for (int c = 0; c < SL_ROW_CELLS; ++c) v8[c] = (i0 + c < hi) ? values[i0 + c - lo] : 0.0;
            return;
        }
        start_row(M, i0);
        const int first = (int)ijk[L];
#pragma unroll
        for (int c = 0; c < SL_ROW_CELLS; ++c) {
            ijk[L] = first + c;
            v8[c] = cell(M);
        }
    }

    // Rounding-error bound of one computed V on this grid, times a safety factor: every computed
    // value is within (2 D + 2) u sum_jk |x_j| |P_jk| |x_k| of the real quadratic form (u = 2^-53);
    // 1e-13 of the same sum with the coordinates replaced by their largest magnitudes on the grid
    // is 90 times that at D = 4.
    SL_HD static double error_margin(const SlDevModel& M) {
        double xmax[D], sum = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) xmax[k] = fmax(fabs(M.m.grid.offset[k]), fabs(M.m.grid.upper[k]));
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < D; ++k) sum += xmax[j] * fabs(M.m.value.matrix[j][k]) * xmax[k];
        return 1e-13 * sum;
    }

    // Are the `ncells` consecutive cells from i0 on (all in ONE row of the last axis) PROVABLY all
    // above vstar - the level the streaming pass of update_safe_set compares with (lyapunov.py:590-606;
    // the level set holds 1.3e5 of the 2.7e8 cells of the 128^4 grid: nearly every row lies far outside
    // it)?  Along the row V is the quadratic c + b t + a t^2 in the last coordinate t.  Where it is
    // strictly monotone over the span - neighbouring cells apart by more than the rounding of two
    // computed values, so that the COMPUTED values are monotone too - its smallest value sits at one
    // end (bounded below by the real-arithmetic value there minus the margin) and its largest at the
    // other; where the row's minimum lies inside the span (a > 0) the bound is the vertex value and
    // the largest cell is the end farther from it, if it beats the other end and its own neighbour by
    // more than the rounding (V is convex: no cell between them is larger than both).  Then only that
    // largest cell is evaluated, exactly, in the reference's order (it is the span's candidate for
    // the range's largest key, lyapunov.py:590-595 when the first cell fails): returns its offset
    // (0 or ncells - 1) with *v_top set; -1: nothing is known, nothing was evaluated.  NaN anywhere
    // makes the comparisons false.  DT > 0, no negation.  Leaves ijk[L] at the span's first cell.
    SL_HD int span_bounded(const SlDevModel& M, int64_t i0, int ncells, double vstar, double margin,
                           double* v_top) {
        start_row(M, i0);
        const int first = (int)ijk[L];
        point(M, L);
        const double t0 = x[L];
        ijk[L] = first + (ncells - 1);
        point(M, L);
        const double t7 = x[L];
        ijk[L] = first;
        const double a = M.m.value.matrix[L][L];
        double b = L > 0 ? lin_pre[L] : 0.0, c = 0.0;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            b += M.m.value.matrix[L][j] * x[j];
            c += lin_pre[j] * x[j];
        }
        const double vt0 = c + t0 * (b + a * t0), vt7 = c + t7 * (b + a * t7);
        const double d0 = 2.0 * a * t0 + b, d7 = 2.0 * a * t7 + b;       // V'(t) at the two ends
        const double h = M.m.grid.unit_maxes[L];
        const bool monotone = (d0 > 0.0 && d7 > 0.0) || (d0 < 0.0 && d7 < 0.0);
        int top = -1;
        double lower = 0.0;
        if (monotone) {
            // least step between neighbouring cells: h (min |V'| - |a| h)
            if (h * (fmin(fabs(d0), fabs(d7)) - fabs(a) * h) > 4.0 * margin) {
                top = d0 > 0.0 ? ncells - 1 : 0;
                lower = fmin(vt0, vt7);
            }
        } else if (a > 0.0 && d0 < 0.0 && d7 > 0.0) {
            // the row's minimum lies inside the span (every whole row of a grid around the origin;
            // one group of eight in sixteen of a 128-cell row): V >= c - b^2 / 4a
            const bool right = vt7 > vt0;
            const double dtop = right ? d7 : -d0;
            if (fabs(vt7 - vt0) > 4.0 * margin && h * (dtop - a * h) > 4.0 * margin) {
                top = right ? ncells - 1 : 0;
                lower = c - (b * b) / (4.0 * a) - margin;
            }
        }
        if (top >= 0 && lower - margin > vstar) {
            ijk[L] = first + top;
            *v_top = cell(M);
            ijk[L] = first;
            return top;
        }
        return -1;
    }

    // eight(), unless span_bounded() proves the eight cells above vstar: returns the position (0 or
    // 7) of the one evaluated cell with v8[position] set; -1: all eight were evaluated.
    SL_HD int eight_bounded(const SlDevModel& M, int64_t i0, double vstar, double margin, double* v8) {
        double vt;
        const int top = span_bounded(M, i0, SL_ROW_CELLS, vstar, margin, &vt);
        if (top >= 0) { v8[top] = vt; return top; }
        const int first = (int)ijk[L];
#pragma unroll
        for (int k = 0; k < SL_ROW_CELLS; ++k) {
            ijk[L] = first + k;
            v8[k] = cell(M);
        }
        return -1;
    }
};
