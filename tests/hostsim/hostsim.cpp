// hostsim.cpp - TEST-ONLY host build of the per-cell arithmetic in safe_learning_amd/csrc/sl_model.h.
//
// The HIP kernels and this file include the very same header; compiling it with g++ lets the CPU
// test-suite check index->state, policy, dynamics, V, L_v, threshold and decrease bit-for-bit
// against the oracle without a GPU.  It is never imported by the product package.
#include <cstdlib>
#include <cstring>
#include "sl_model.h"

static void make_model(const sl_model_desc* desc, SlDevModel* M) {
    std::memset(M, 0, sizeof(*M));
    M->m = *desc;
    SlGridFast& gf = M->gf;
    gf.d = desc->grid.d;
    gf.all_pow2 = 1;
    gf.nindex = 1;
    for (int k = 0; k < gf.d; ++k) {
        int64_t n = desc->grid.num_points[k];
        gf.nindex *= n;
        gf.num32[k] = (uint32_t)n;
        if ((n & (n - 1)) == 0) { int s = 0; while ((1ll << s) < n) ++s; gf.shift[k] = s; }
        else gf.all_pow2 = 0;
    }
    M->in_dim = desc->grid.d + desc->policy.m;
    M->uncertain = 0;
    const int lk = desc->lipschitz.lv_kind;
    M->m.lipschitz.lv_cols = (lk == SL_LIP_CONST || lk == SL_LIP_NORM_LINEAR || lk == SL_LIP_NORM_GRAD) ? 1 : desc->grid.d;
}

extern "C" {

// deterministic-dynamics decrease check of cells [lo, hi): quadratic V, closed-form policy
int hs_det_cells(const sl_model_desc* desc, int64_t lo, int64_t hi, double* values,
                 uint8_t* negative, double* dbg) {
    SlDevModel M;
    make_model(desc, &M);
    const SlDims n = sl_dims<0, 0>(M);
    const int d = n.d;
    for (int64_t idx = lo; idx < hi; ++idx) {
        double x[SL_P], u[SL_M], nxt[SL_D], lv_x[SL_D], lv_n[SL_D], err[SL_D];
        sl_index_to_state(M.m.grid, M.gf, d, idx, x);
        sl_policy_closed_form(M, n, x, u);
        sl_append_action(n, u, x);
        sl_dynamics_det<0>(M, n, x, nxt);
        const double v_x = sl_quadratic(M.m.value, d, x);
        const double v_n = sl_quadratic(M.m.value, d, nxt);
        const double dec = sl_decrease(M, d, v_x, v_n, lv_n, err);
        sl_lv(M, d, x, lv_x);
        const double thr = sl_threshold(M, d, lv_x, M.m.lipschitz.tau, x);
        values[idx - lo] = v_x;
        negative[idx - lo] = dec < thr ? 1 : 0;
        if (dbg) {
            double* o = dbg + (idx - lo) * (2 + 2 * d);
            o[0] = dec; o[1] = thr;
            for (int k = 0; k < d; ++k) { o[2 + k] = nxt[k]; o[2 + d + k] = 0.0; }
        }
    }
    return 0;
}

// piecewise-linear interpolation at explicit points
int hs_tri_eval(const sl_grid_desc* grid, int nsimplex, const int32_t* simplices,
                const double* hyper, const double* discrete_points, int project, int ncols,
                const double* table, int64_t npts, const double* pts, int col, double* out,
                double* grad) {
    static SlTri t;
    std::memset(&t, 0, sizeof(t));
    t.grid = *grid;
    t.nsimplex = nsimplex; t.project = project; t.ncols = ncols; t.set = 1;
    const int d = grid->d;
    for (int s = 0; s < nsimplex; ++s) {
        for (int v = 0; v <= d; ++v) t.simplices[s][v] = simplices[s * (d + 1) + v];
        for (int k = 0; k < d; ++k)
            for (int j = 0; j < d; ++j) t.hyper[s][k][j] = hyper[(s * d + k) * d + j];
    }
    int64_t stride = 1, total = 0;
    for (int k = d - 1; k >= 0; --k) { t.stride[k] = stride; stride *= grid->num_points[k]; }
    for (int k = 0; k < d; ++k) { t.points_off[k] = (int32_t)total; total += grid->num_points[k]; }
    t.points = discrete_points;
    t.table = table;
    sl_tri_finish(t, discrete_points);
    if (getenv("SL_HOSTSIM_LOAD_POINTS")) t.affine_points = 0;
    if (getenv("SL_HOSTSIM_NO_REGIONS")) { std::memset(t.ncand, 0, sizeof(t.ncand)); t.has_fine = 0; }
    if (col == -2) {             // the candidate table of the point location shortcut
        for (int c = 0; c < SL_TRI_CODES; ++c) out[c] = (double)t.ncand[c];
        return 0;
    }
    if (col == -3) {             // 1.0 where the candidate phase of the point location resolves the point
        for (int64_t i = 0; i < npts; ++i) {
            const double* x = pts + i * d;
            double z[4], zn[4], best_min = -1e300;
            for (int k = 0; k < d; ++k) {
                double c = x[k] - t.grid.offset[k];
                const double eps2 = 2.0 * 2.220446049250313e-16;
                const double lo = 0.0 + eps2, hi = (t.grid.upper[k] - t.grid.offset[k]) - eps2;
                c = (c < lo) ? lo : c;
                c = (c > hi) ? hi : c;
                z[k] = sl_fmod_exact(c, t.grid.unit_maxes[k], t.inv_unit[k]);
                zn[k] = z[k] * t.inv_unit[k];
            }
            const int code = sl_tri_region_code(d, zn);
            int list[SL_TRI_MAXCAND], n = 0;
            bool fine = false;
            if (d == 4 && t.has_fine) {
                const uint8_t* f = t.fine[t.perm_index[code] * 64 + sl_tri_sum_code(d, zn)];
                for (int q = 0; q < 4 && f[q] != 0xFF; ++q) list[n++] = f[q];
                fine = n > 0;
            }
            if (!fine) for (int q = 0; q < t.ncand[code]; ++q) list[n++] = t.cand[code][q];
            for (int q = 0; q < n; ++q) {
                const int s = list[q];
                double w0 = 1.0, wmin = 1e300;
                for (int j = 0; j < d; ++j) {
                    double w = -t.hyper_c[s][j];
                    for (int k = 0; k < d; ++k) w = fma(z[k], t.hyper[s][k][j], w);
                    w0 -= w;
                    wmin = fmin(wmin, w);
                }
                wmin = fmin(wmin, w0);
                if (wmin > best_min) best_min = wmin;
            }
            out[i] = (best_min > 0.0 ? 1.0 : 0.0) + (fine ? 2.0 : 0.0) + 4.0 * n;
        }
        return 0;
    }
    if (col == -4) {             // rectangle index per point (digitize semantics, functions.py:1116-1124)
        for (int64_t i = 0; i < npts; ++i) {
            double corner = 0.0, mul = 1.0;
            for (int k = d - 1; k >= 0; --k) {
                corner += mul * (double)sl_rectangle_1d(t, k, pts[i * d + k]);
                mul *= (double)(grid->num_points[k] - 1);
            }
            out[i] = corner;
        }
        return 0;
    }
    if (col == -1) {             // the Bellman sweeps' lookup (column 0, compile-time dimension)
        for (int64_t i = 0; i < npts; ++i) {
            const double* x = pts + i * d;
            out[i] = d == 1 ? sl_tri_value_fast<1>(t, x) : d == 2 ? sl_tri_value_fast<2>(t, x)
                   : d == 3 ? sl_tri_value_fast<3>(t, x) : d == 4 ? sl_tri_value_fast<4>(t, x)
                   : sl_tri_value_fast<0>(t, x);
        }
        return 0;
    }
    for (int64_t i = 0; i < npts; ++i)
        out[i] = sl_tri_eval(t, pts + i * d, col, grad ? grad + i * d : nullptr);
    return 0;
}

double hs_fmod_exact(double a, double b) { return sl_fmod_exact(a, b); }
double hs_fmod_exact_inv(double a, double b) { return sl_fmod_exact(a, b, 1.0 / b); }
double hs_exp_nonpos(double x) { return sl_exp_nonpos(x); }
void hs_sincos(double x, double* s, double* c) { sl_sincos(x, s, c); }

uint64_t hs_vbits(double v) { return sl_vbits(v); }
uint64_t hs_vbits_fast(double v) { return sl_vbits_fast(v); }
int hs_values_implicit_ok(const sl_model_desc* desc) { return sl_values_implicit_ok(*desc) ? 1 : 0; }

// the ordering keys as the streaming passes recompute them (sl_level.hip: SlRowValues<D>::eight, a
// thread's 8 cells of a row) and as k_values writes them (sl_quadratic on the np.linspace points)
int hs_row_values(const sl_model_desc* desc, int64_t n, double* rows, double* points) {
    SlDevModel M;
    make_model(desc, &M);
    const int d = M.m.grid.d;
    if (n % SL_ROW_CELLS) return -1;
    for (int64_t i0 = 0; i0 < n; i0 += SL_ROW_CELLS) {
        double v8[SL_ROW_CELLS];
        switch (d) {
            case 1: { SlRowValues<1> r; r.eight(M, nullptr, 0, n, i0, v8); break; }
            case 2: { SlRowValues<2> r; r.eight(M, nullptr, 0, n, i0, v8); break; }
            case 3: { SlRowValues<3> r; r.eight(M, nullptr, 0, n, i0, v8); break; }
            case 4: { SlRowValues<4> r; r.eight(M, nullptr, 0, n, i0, v8); break; }
            default: return -2;
        }
        for (int c = 0; c < SL_ROW_CELLS; ++c) rows[i0 + c] = v8[c];
    }
    for (int64_t i = 0; i < n; ++i) {
        double x[SL_P];
        sl_index_to_grid_point(M.m.grid, M.gf, d, i, x);
        points[i] = sl_quadratic(M.m.value, d, x);
    }
    return 0;
}
// the bounded variant of the streaming pass (SlRowValues<D>::eight_bounded): per group of 8 cells
// top[g] = -1 (all evaluated) or the position of the one evaluated cell, vals[8 g + c] its values
int hs_row_bounded(const sl_model_desc* desc, int64_t n, double vstar, int32_t* top, double* vals,
                   double* margin_out) {
    SlDevModel M;
    make_model(desc, &M);
    const int d = M.m.grid.d;
    if (n % SL_ROW_CELLS) return -1;
    for (int64_t i0 = 0; i0 < n; i0 += SL_ROW_CELLS) {
        double v8[SL_ROW_CELLS];
        for (int c = 0; c < SL_ROW_CELLS; ++c) v8[c] = 0.0;
        int t = -2;
        switch (d) {
            case 1: { SlRowValues<1> r; *margin_out = r.error_margin(M); t = r.eight_bounded(M, i0, vstar, *margin_out, v8); break; }
            case 2: { SlRowValues<2> r; *margin_out = r.error_margin(M); t = r.eight_bounded(M, i0, vstar, *margin_out, v8); break; }
            case 3: { SlRowValues<3> r; *margin_out = r.error_margin(M); t = r.eight_bounded(M, i0, vstar, *margin_out, v8); break; }
            case 4: { SlRowValues<4> r; *margin_out = r.error_margin(M); t = r.eight_bounded(M, i0, vstar, *margin_out, v8); break; }
            default: return -2;
        }
        top[i0 / SL_ROW_CELLS] = t;
        for (int c = 0; c < SL_ROW_CELLS; ++c) vals[i0 + c] = v8[c];
    }
    return 0;
}
// SlRowValues::span_bounded over spans of `ncells` cells (ncells divides the last axis): the offset
// of the evaluated cell (or -1) and its value per span
int hs_span_bounded(const sl_model_desc* desc, int64_t n, int ncells, double vstar, int32_t* top,
                    double* vtop) {
    SlDevModel M;
    make_model(desc, &M);
    const int d = M.m.grid.d;
    if (ncells < 1 || n % ncells || M.m.grid.num_points[d - 1] % ncells) return -1;
    for (int64_t i0 = 0; i0 < n; i0 += ncells) {
        double v = 0.0;
        int t = -2;
        switch (d) {
            case 1: { SlRowValues<1> r; t = r.span_bounded(M, i0, ncells, vstar, r.error_margin(M), &v); break; }
            case 2: { SlRowValues<2> r; t = r.span_bounded(M, i0, ncells, vstar, r.error_margin(M), &v); break; }
            case 3: { SlRowValues<3> r; t = r.span_bounded(M, i0, ncells, vstar, r.error_margin(M), &v); break; }
            case 4: { SlRowValues<4> r; t = r.span_bounded(M, i0, ncells, vstar, r.error_margin(M), &v); break; }
            default: return -2;
        }
        top[i0 / ncells] = t;
        vtop[i0 / ncells] = v;
    }
    return 0;
}
double hs_vbits_to_double(uint64_t b) { return sl_vbits_to_double(b); }


// sum-of-products kernels (sl_gp_set_head_kernel): K[i][j] = k(a_i, b_j) and the diagonal k(a_i, a_i)
// exactly as the sweep kernels evaluate them (sl_kernel_eval / sl_kernel_diag of sl_model.h)
int hs_kernel_matrix(const sl_gp_kernel* ks, int p, const double* a, int na, const double* b, int nb,
                     double* out, double* diag) {
    for (int i = 0; i < na; ++i) {
        double xa[SL_P] = {};
        for (int q = 0; q < p; ++q) xa[q] = a[i * p + q];
        for (int j = 0; j < nb; ++j) {
            double xb[SL_P] = {};
            for (int q = 0; q < p; ++q) xb[q] = b[j * p + q];
            out[i * nb + j] = sl_kernel_eval(*ks, p, xa, xb);
        }
        diag[i] = sl_kernel_diag(*ks, p, xa);
    }
    return 0;
}
// sl_kernel_eval2 (two training points in one walk over the factors - what k_gp_small evaluates)
// against sl_kernel_eval: number of pairs whose values differ in any bit
int hs_kernel_eval2_mismatches(const sl_gp_kernel* ks, int p, const double* a, int na, const double* b, int nb) {
    int bad = 0;
    for (int i = 0; i + 1 < na; i += 2) {
        double x0[SL_P] = {}, x1[SL_P] = {};
        for (int q = 0; q < p; ++q) { x0[q] = a[i * p + q]; x1[q] = a[(i + 1) * p + q]; }
        for (int j = 0; j < nb; ++j) {
            double xb[SL_P] = {};
            for (int q = 0; q < p; ++q) xb[q] = b[j * p + q];
            double k0, k1;
            sl_kernel_eval2(*ks, p, x0, x1, xb, &k0, &k1);
            const double e0 = sl_kernel_eval(*ks, p, x0, xb), e1 = sl_kernel_eval(*ks, p, x1, xb);
            if (memcmp(&k0, &e0, 8) || memcmp(&k1, &e1, 8)) ++bad;
        }
    }
    return bad;
}
}  // extern "C"
