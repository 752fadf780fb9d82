// sl_gp_small.hip - the GP-dynamics Lyapunov sweep for SMALL training sets (n_pad <= 256 per head):
// the shape of the reference's own notebooks (<= 130 points on 3-9 M-cell 2-D grids with a table
// value function and a table policy, examples/inverted_pendulum.ipynb:112, 152-177).
//
// k_gp_sweep (sl_gp.hip) treats such a model like a large one - a workgroup per 16-cell tile, k_x
// through LDS, three workgroup barriers per tile - and spends 13 us per tile on 0.3 MFLOP.  Here
// nothing is shared between wavefronts but read-only tables:
//
//  * a WAVEFRONT owns a tile of 64 consecutive cells and never meets a barrier after the staging
//    of the workgroup: eight wavefronts per workgroup (two per SIMD) run at their own pace, one
//    covering the other's latencies (table lookups, LDS round trips, exponentials);
//  * k_x is produced in registers, directly in the B-fragment layout of v_mfma_f64_16x16x4_f64
//    (lane = (training point k of the slab, cell)): one exponential per (point, cell), no LDS
//    transpose.  With n <= 256 the exponentials are 10-20 % of the GEMM's time;
//  * the lower triangle of Linv (MFMA A-fragments, 72 KB at n_pad = 128) sits in LDS next to the
//    scaled training inputs and alpha'; for 128 < n_pad <= 256 the fragments stay in L2;
//  * a = Linv k_x for 16 cells x 128 rows at a time: eight accumulator tiles per wavefront, every
//    k_x fragment feeds all the row blocks below its diagonal;
//  * a FunctionStack of single-output GPs (the notebooks' dynamics model: one GP per state
//    dimension, own kernel and training inputs, functions.py:278-291) is a loop over the heads
//    inside the tile: each head has its own scaled inputs, alpha' and factor;
//  * the decrease check runs one cell per lane on all 64 lanes (the table flavours cost thousands
//    of cycles per cell: value table at x and at the posterior mean, its gradient, the policy
//    table), the 64-lane ballot is the tile's mask word.
//
// Same results as k_gp_sweep (same operation order per cell up to the order of the k_x . alpha'
// and |a|^2 sums, i.e. rounding); tests/test_gpu_lyapunov.py runs both on the same models.
#include "sl_common.h"

typedef double sl_d4 __attribute__((ext_vector_type(4)));
typedef double sl_d2 __attribute__((ext_vector_type(2)));

namespace gps {
// wavefronts per workgroup: 8 (two per SIMD), or 12 - three per SIMD, which caps the kernel at 168
// registers (a handful of spills in the table check of the general flavour) and hides more of the
// latency of its table look-ups; taken when the per-wavefront scratch of 12 wavefronts still fits LDS;
// or 16 - four per SIMD at 128 registers, the factor read from L2 - on large sweeps (launch_small)
constexpr int WAVES_MIN = 8, WAVES_MAX = 12, WAVES_TOP = 16;
// ... or 4 on sweeps with fewer tiles than a device of 8-wavefront workgroups has wavefronts (the
// notebooks' coarse 251 x 251 grids: 985 tiles): the workgroups then reach every CU (launch_small)
constexpr int WAVES_TINY = 4;
// (a 10-wavefront workgroup - what fits beside a factor in LDS - measured SLOWER than 8 on the
// single-head table sweep: 2.31 against 2.22 ms, profiles/r04_configs.jsonl vs r04_gp_small_waves_ab.txt)
constexpr int PRB = 8;            // row blocks (of 16 rows) per pass: 128 rows
// fragment pairs (1 KiB each: two slabs of 4 training points x 16 rows) of the lower triangle in
// front of row block I: row block i needs the slab pairs 0 .. 2 i + 1
__host__ __device__ constexpr int tri_offset(int I) { return I * (I + 1); }
// LDS doubles of a head's kernel description (sum-of-products heads only)
constexpr int KERNEL_DOUBLES = (int)((sizeof(sl_gp_kernel) + 15) / 16) * 2;
}  // namespace gps

// ALDS: the A fragments are read from the workgroup's LDS copy (else from L2).
// KERN: some head carries a sum-of-products kernel (sl_gp_set_head_kernel); the plain-RBF
// instantiation has no trace of that path in its generation loop.
template <bool GENERAL, int DT, int MT, bool ALDS, int WAVES, bool KERN>
__global__ __launch_bounds__(64 * WAVES) void k_gp_small(
    const SlDevModel M, const SlGpDev gp, SlAux aux_arg, int64_t lo, int64_t hi, int64_t ntiles,
    const uint64_t* __restrict__ init_bits, const double* __restrict__ values,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    int head_doubles, const double* __restrict__ points, unsigned long long* __restrict__ ticket,
    int diag) {
    using namespace gps;
#ifdef SL_DIAG
    const int dg = diag;         // development builds (tools/build_variant.sh): phases switched off for timing attribution -
                                 // 1 no kernel evaluation, 2 no MFMAs, 4 no fragment loads, 8 no check, 16 no policy, 32 no GEMM passes
#else
    constexpr int dg = 0;
    (void)diag;
#endif
    __shared__ SlTriLds<GENERAL> tri_lds;
    __shared__ uint64_t sv[WAVES];
    __shared__ int64_t si[WAVES];
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlDims nd = sl_dims<DT, MT>(M);
    const int d = nd.d, p = nd.p;
    const int nheads = gp.nheads;
    // LDS: per head [xs p x n_pad | alpha' n_pad x dout | lower-triangle fragments (ALDS)], then the
    // wavefronts' scratch: ssq [64], mean [64][d], err [64][d] (the scaled inputs of a cell block
    // travel between lanes by ds_bpermute: no LDS copy - 1.5 KB per wavefront at p = 3, which is what
    // lets the factor of a 128-point table model sit in LDS beside sixteen wavefronts)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 15, lk = lane >> 4;
    const int wstride = 64 * (1 + 2 * d);
    double* ssq_w = smem + head_doubles + wave * wstride;
    double* mean_w = ssq_w + 64;
    double* err_w = mean_w + 64 * d;

    // ---- staging (once per workgroup) -----------------------------------------------------------
    {
        double* dst = smem;
        for (int h = 0; h < nheads; ++h) {
            const SlGpHeadDev& hd = gp.head[h];
            // row blocks that hold training points: the padding of the upload (a multiple of the
            // 16x16x4 fallback's 64-row panels) is rows and columns of zeros - neither staged
            // nor multiplied here (a 130-point head of the notebooks works on 144 of its 192)
            const int n_pad = hd.n_pad, nrb = (hd.n + 15) / 16;
            for (int k = tid; k < p * n_pad; k += 64 * WAVES) dst[k] = hd.xs[k];
            dst += (p * n_pad + 1) & ~1;
            for (int k = tid; k < n_pad * hd.dout; k += 64 * WAVES) dst[k] = hd.alpha[k];
            dst += (n_pad * hd.dout + 1) & ~1;
            if (KERN && hd.kernel) {
                // the kernel description: read at every slab pair of every tile
                const double* src = reinterpret_cast<const double*>(hd.kernel);
                for (int k = tid; k < (int)(sizeof(sl_gp_kernel) / sizeof(double)); k += 64 * WAVES) dst[k] = src[k];
                dst += KERNEL_DOUBLES;
            }
            if (ALDS) {
                // row block I, slab pair s2 (s2 <= 2 I + 1): 128 doubles at tri_offset(I) + s2
                for (int I = 0; I < nrb; ++I) {
                    const double* src = hd.mpack + (size_t)I * hd.nslab2 * 128;
                    double* to = dst + (size_t)tri_offset(I) * 128;
                    for (int k = tid; k < (2 * I + 2) * 128; k += 64 * WAVES) to[k] = src[k];
                }
                dst += (size_t)tri_offset(nrb) * 128;
            }
        }
    }
    __syncthreads();

    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;

    // Tiles are drawn from a counter, one per wavefront: table look-ups and saturation kinks make
    // tiles differ in cost (3 - 7 % on the 3e6-cell configurations, profiles/r04_gp_small_tickets_ab.txt).
    // Small sweeps (ticket == nullptr: a handful of tiles per wavefront) keep the fixed list - there
    // the counter's memset and atomics cost more than they balance.
    auto next_tile = [&](int64_t previous) -> int64_t {
        if (!ticket) return previous < 0 ? (int64_t)blockIdx.x * WAVES + wave : previous + (int64_t)gridDim.x * WAVES;
        int64_t t = 0;
        if (lane == 0) t = (int64_t)atomicAdd(ticket, 1ull);
        return ((int64_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    for (int64_t tile = next_tile(-1); tile < ntiles; tile = next_tile(tile)) {
        const int64_t tile_base = lo + tile * 64;
        const int64_t idx = tile_base + lane;
        const bool valid = idx < hi;
        const int64_t gidx = valid ? idx : hi - 1;
        // ---- GP input [x, policy(x)] of this lane's cell (kept for the check) ----------------------
        double x[SL_P], u[SL_M];
        sl_cell_state(M, d, gidx, points, x);
        if (dg & 16) { for (int a = 0; a < SL_M; ++a) u[a] = 0.0; } else
        sl_policy_any<GENERAL>(M, nd, aux.tri, gidx, x, u);
        sl_append_action(nd, u, x);

        const double* head_base = smem;
        for (int h = 0; h < nheads; ++h) {
        const SlGpHeadDev& hd = gp.head[h];
        const int n_pad = hd.n_pad, dout = hd.dout, nslab2 = hd.nslab2, col0 = hd.col0;
        const int nrb = (hd.n + 15) / 16, npass = (nrb + PRB - 1) / PRB;
        const double variance = hd.variance;
        const double* xs_l = head_base;                              // [p][n_pad]
        const double* alpha_l = xs_l + ((p * n_pad + 1) & ~1);       // [n_pad][dout]
        const double* kern_l = alpha_l + ((n_pad * dout + 1) & ~1);  // kernel description (KERN heads)
        // null: the RBF of sl_gp_set_head
        const sl_gp_kernel* kern = (KERN && hd.kernel) ? reinterpret_cast<const sl_gp_kernel*>(kern_l) : nullptr;
        const double* a_l = kern_l + (kern ? KERNEL_DOUBLES : 0);    // lower-triangle fragments (ALDS)
        head_base = a_l + (ALDS ? (size_t)tri_offset(nrb) * 128 : 0);

        for (int pass = 0; pass < ((dg & 32) ? 0 : npass); ++pass) {
            // The SHORT pass comes first: a pass regenerates k_x up to its diagonal, and the top of the
            // triangle needs the fewest slab pairs (130 points, nine row blocks: 2 + 18 slab pairs with
            // the one-block pass in front, 16 + 18 with it behind).
            const int first = nrb - (npass - 1) * PRB;          // row blocks of pass 0 (1 .. PRB)
            const int rb0 = pass == 0 ? 0 : first + (pass - 1) * PRB;   // first row block of the pass
            const int rbn = pass == 0 ? first : PRB;            // row blocks in it
            const int ns2 = 2 * (rb0 + rbn);                    // slab pairs up to its diagonal
            const int new_s2 = 2 * rb0;                         // training points not seen before
            for (int cb = 0; cb < 4; ++cb) {
                double xg[SL_P];
#pragma unroll
                for (int q = 0; q < SL_P; ++q)      // this head's scaled inputs (own lengthscales) of cell 16 cb + lcol
                    xg[q] = (q < p) ? __shfl(x[q] * hd.inv_ls[q], 16 * cb + lcol, 64) : 0.0;
                sl_d4 acc[PRB];
#pragma unroll
                for (int r = 0; r < PRB; ++r) acc[r] = (sl_d4){0.0, 0.0, 0.0, 0.0};
                double gm[SL_D];
#pragma unroll
                for (int dd = 0; dd < SL_D; ++dd) gm[dd] = 0.0;

                for (int s2 = 0; s2 < ns2; ++s2) {
                    // k_x of the slab pair, B-fragment layout: lane (k, cell) holds the points
                    // 8 s2 + k (first slab) and 8 s2 + 4 + k (second slab)
                    const int j0 = 8 * s2 + lk, j1 = j0 + 4;
                    double z0 = 0.0, z1 = 0.0;
#pragma unroll
                    for (int q = 0; q < SL_P; ++q) {
                        if (q < p) {
                            const double d0 = xs_l[q * n_pad + j0] - xg[q];
                            const double d1 = xs_l[q * n_pad + j1] - xg[q];
                            z0 = fma(d0, d0, z0);
                            z1 = fma(d1, d1, z1);
                        }
                    }
                    double k0, k1;
                    if (dg & 1) {
                        k0 = z0; k1 = z1;
                    } else if (kern) {                          // sum-of-products kernel, unscaled inputs
                        double xa[SL_P], xb[SL_P];
#pragma unroll
                        for (int q = 0; q < SL_P; ++q) {
                            xa[q] = (q < p) ? xs_l[q * n_pad + j0] : 0.0;
                            xb[q] = (q < p) ? xs_l[q * n_pad + j1] : 0.0;
                        }
                        // the padding columns of the tables are zero; their fragments of the
                        // factor and of alpha' are zero as well
                        sl_kernel_eval2(*kern, p, xa, xb, xg, &k0, &k1);
                    } else {
                        k0 = variance * sl_exp_nonpos(-0.5 * z0);
                        k1 = variance * sl_exp_nonpos(-0.5 * z1);
                    }
                    if (s2 >= new_s2) {                         // posterior mean k_x . alpha'
#pragma unroll
                        for (int dd = 0; dd < SL_D; ++dd) {
                            if (dd < dout) {
                                gm[dd] = fma(k0, alpha_l[j0 * dout + dd], gm[dd]);
                                gm[dd] = fma(k1, alpha_l[j1 * dout + dd], gm[dd]);
                            }
                        }
                    }
                    // every row block on or below the diagonal of this slab pair
                    const int rmin = (s2 >> 1) - rb0;           // first active row block of the pass
#pragma unroll
                    for (int r = 0; r < PRB; ++r) {
                        if (r >= rmin && r < rbn) {
                            const int I = rb0 + r;
                            sl_d2 a;
                            if (dg & 4)
                                a = (sl_d2){1.0, 1.0};
                            else if (ALDS)
                                a = *reinterpret_cast<const sl_d2*>(a_l + (size_t)(tri_offset(I) + s2) * 128 + lane * 2);
                            else
                                a = *reinterpret_cast<const sl_d2*>(hd.mpack + ((size_t)I * nslab2 + s2) * 128 + lane * 2);
                            if (dg & 2) { acc[r].x += a.x * k0; acc[r].y += a.y * k1; continue; }
                            acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, k0, acc[r], 0, 0, 0);
                            acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, k1, acc[r], 0, 0, 0);
                        }
                    }
                }
                // |a|^2 of the pass's rows for this lane's cell, folded over the four lane groups
                double ss = 0.0;
#pragma unroll
                for (int r = 0; r < PRB; ++r) {
                    const sl_d4 t = acc[r];
                    ss = fma(t.x, t.x, ss);
                    ss = fma(t.y, t.y, ss);
                    ss = fma(t.z, t.z, ss);
                    ss = fma(t.w, t.w, ss);
                }
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
#pragma unroll
                for (int dd = 0; dd < SL_D; ++dd) {
                    if (dd < dout) {
                        gm[dd] += __shfl_xor(gm[dd], 16, 64);
                        gm[dd] += __shfl_xor(gm[dd], 32, 64);
                    }
                }
                if (lane < 16) {
                    const int c = 16 * cb + lane;
                    ssq_w[c] = (pass == 0 ? 0.0 : ssq_w[c]) + ss;
#pragma unroll
                    for (int dd = 0; dd < SL_D; ++dd)
                        if (dd < dout)
                            mean_w[c * d + col0 + dd] = (pass == 0 ? 0.0 : mean_w[c * d + col0 + dd]) + gm[dd];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        {   // this head's error bound for its output columns (lane = cell)
            const double prior = kern ? sl_kernel_diag(*kern, p, x) : variance;   // functions.py:450
            const double var = prior - ssq_w[lane];                        // functions.py:451
            const double e = gp.beta * sqrt(var);                          // functions.py:514
#pragma unroll
            for (int dd = 0; dd < SL_D; ++dd)
                if (dd < dout) err_w[lane * d + col0 + dd] = e;
        }
        }   // heads
        __builtin_amdgcn_wave_barrier();

        // ---- per-cell decrease check (lane = cell), mask word, failing-cell key ----------------------
        bool negative = false;
        double v_x = 0.0;
        {
            double prior[SL_D], mean[SL_D], err[SL_D];
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);   // m(x*), functions.py:439
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d) {
                    mean[k] = mean_w[lane * d + k] + prior[k];
                    err[k] = err_w[lane * d + k];
                }
            }
            if (valid && !(dg & 8)) {
                SlCellCheck c = sl_cell_check<SlSweepFlavour<GENERAL>::value>(M, d, aux, x, mean, err);
                negative = c.negative;
                v_x = values ? values[idx - lo] : c.v_x;               // ordering key: lyapunov.py:512
                if (dbg) {
                    double* o = dbg + (idx - lo) * (2 + 2 * d);
                    o[0] = c.decrease; o[1] = c.threshold;
#pragma unroll
                    for (int k = 0; k < SL_D; ++k) if (k < d) { o[2 + k] = mean[k]; o[2 + d + k] = err[k]; }
                }
            }
        }
        const uint64_t word = __ballot(negative);
        uint64_t init = 0ull;
        if (init_bits) init = init_bits[(tile_base - lo) >> 6];
        if (lane == 0) neg_bits[(tile_base - lo) >> 6] = word;
        const bool ok = negative || ((init >> lane) & 1ull);
        if (valid && !ok) sl_key_min(best_v, best_i, sl_vbits(v_x), idx);
        __builtin_amdgcn_wave_barrier();      // the scratch is rewritten by the next tile
    }
    __syncthreads();
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (tid == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// =============================================================================================
// host side
// =============================================================================================
// LDS doubles of a launch without the factor's lower triangle: per head the scaled inputs and
// alpha', plus the per-wavefront scratch of the check
static size_t small_lds_doubles(sl_ctx* ctx, int p, int d) {
    size_t small = 0;
    for (int k = 0; k < ctx->h_gp.nheads; ++k) {
        const SlGpHeadHost& h = ctx->gp_heads[k];
        small += (size_t)((p * h.n_pad + 1) & ~1) + (size_t)((h.n_pad * h.dout + 1) & ~1) +
                 (h.d_kernel ? gps::KERNEL_DOUBLES : 0);
    }
    return small + (size_t)gps::WAVES_MIN * 64 * (1 + 2 * d);
}

static size_t lds_capacity(bool general) {
    return 160 * 1024 - (general ? sizeof(SlTriLds<true>) : 0) - 256;
}

// models the kernel takes: every head with a padded capacity of at most 256 training points, and
// the heads' inputs / alpha' / the check's scratch fit LDS (a 6-D stack of six 256-point heads
// does not: it stays on k_gp_sweep)
bool sl_gp_small_supports(sl_ctx* ctx, const SlDevModel& model) {
    if (ctx->env.gp_small == 0) return false;
    if (ctx->h_gp.nheads < 1) return false;
    for (int k = 0; k < ctx->h_gp.nheads; ++k) {
        const SlGpHeadHost& h = ctx->gp_heads[k];
        if (!h.set || h.n_pad > 256 || h.n_pad % 16 || h.p != model.in_dim) return false;
    }
    return sizeof(double) * small_lds_doubles(ctx, model.in_dim, model.m.grid.d)
           <= lds_capacity(sl_model_is_general(model));
}

template <bool GENERAL, int DT, int MT>
static int launch_small(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                        const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                        int* nblocks, double* d_dbg, const double* d_points) {
    using namespace gps;
    const int p = model.in_dim, d = model.m.grid.d;
    const int64_t ntiles = (hi - lo + 63) / 64;
    // per head: scaled inputs, alpha' and (when everything fits) the factor's lower triangle
    size_t small = 0, tri = 0;
    for (int k = 0; k < ctx->h_gp.nheads; ++k) {
        const SlGpHeadHost& h = ctx->gp_heads[k];
        small += (size_t)((p * h.n_pad + 1) & ~1) + (size_t)((h.n_pad * h.dout + 1) & ~1) +
                 (h.d_kernel ? KERNEL_DOUBLES : 0);
        tri += (size_t)tri_offset((h.n + 15) / 16) * 128;         // row blocks with training points
    }
    const size_t cap = lds_capacity(GENERAL);
    auto scratch_of = [&](int waves) { return (size_t)waves * 64 * (1 + 2 * d); };
    auto fits = [&](int waves, bool with_tri) {
        return sizeof(double) * (small + (with_tri ? tri : 0) + scratch_of(waves)) <= cap;
    };
    // the factor in LDS matters most, then the third wavefront per SIMD
    const int wmax = ctx->env.gp_small_waves >= 0 ? ctx->env.gp_small_waves : WAVES_TOP;
    int waves = WAVES_MIN;
    bool alds = fits(WAVES_MIN, true);
    // (the runtime-dimension instantiations need all 256 registers: two wavefronts per SIMD)
    const int wcap = ctx->env.gp_small_waves;        // SL_GP_SMALL_WAVES: 8 = eight only, 12 = no sixteen, 4 = four or eight
    auto allowed = [&](int w) {
        return w == WAVES_MIN || wcap < 0 || (w > WAVES_MIN && w <= wcap) || (w == WAVES_TINY && wcap == WAVES_TINY);
    };
    if (DT > 0 && ntiles < (int64_t)4 * ctx->num_cu * WAVES_TOP) {
        // Below four rounds of the largest workgroups a sweep is a handful of ROUNDS of one tile per
        // wavefront, and a round lasts as long as its slowest tile: the workgroup size with the least
        // (rounds x time of a full round) - 0.105 / 0.128 / 0.16 / 0.22 ms for 4 / 8 / 12 / 16 wavefronts
        // per workgroup on the 128-point table model (0.19 on sixteen when the factor comes from L2
        // anyway: more wavefronts cover more of that latency), not rounded up where the tile counter is
        // on (more than four tiles per wavefront: 12 688 tiles take 0.66 ms on twelve, 0.79 on
        // sixteen).  Measured (profiles/r06_gp_small_lds_ab.txt; tiles: ms before -> after): 1 416:
        // 0.203 -> 0.133, 2 513: 0.247 -> 0.167, 3 922: 0.305 -> 0.245, 7 678: 0.455 -> 0.429; two
        // notebook heads on 2 513: 0.706 -> 0.445; 985 tiles on four wavefronts 0.113 (0.128 on eight;
        // 0.131 with the factor read from L2 instead of staged in LDS by every workgroup).
        const int sizes[4] = {WAVES_TINY, WAVES_MIN, WAVES_MAX, WAVES_TOP};
        // (sixteen WITHOUT the factor in LDS where twelve would have it: the large-sweep rule below
        // keeps twelve in that case - 2.28 against 2.11 ms - and so does the weight)
        const int weight[4] = {105, 128, 160, fits(WAVES_TOP, true) ? 220 : (fits(WAVES_MAX, true) ? 240 : 190)};
        int64_t best = -1;
        for (int k = 0; k < 4; ++k) {
            if (!allowed(sizes[k]) || !fits(sizes[k], false)) continue;
            const int64_t per_round = (int64_t)sizes[k] * ctx->num_cu;
            // (eight wavefronts with the counter on: 10 026 tiles in 0.61 ms against 0.595 in four rounds of twelve)
            const int64_t cost = ntiles > 4 * per_round ? ntiles * (weight[k] + (k == 1 ? 12 : 0)) / per_round
                                                        : ((ntiles + per_round - 1) / per_round) * weight[k];
            if (best < 0 || cost < best) { best = cost; waves = sizes[k]; }
        }
        alds = fits(waves, true);
    } else if (DT > 0 && allowed(WAVES_MAX) && fits(WAVES_MAX, alds)) {
        waves = WAVES_MAX;
    }
    // Large sweeps: FOUR wavefronts per SIMD (sixteen per workgroup, 128 registers - the spills that
    // costs land in the per-cell check, none in the slab-pair loop): a wavefront's time per tile is a
    // sum of latencies that only other wavefronts cover.  With the factor in LDS if that still fits;
    // else with the factor read from L2, which beats two wavefronts per SIMD with it in LDS and three
    // without (2001 x 1501 cells, 128 points, table flavour: 2.00 against 2.18 ms; two heads 3.41
    // against 3.55; 48^4 cells, 192 points: 6.19 against 6.95) but not three with it in LDS (2048^2
    // cells, 128 points: 2.28 against 2.11).  Below four tiles per wavefront of a full device the
    // smaller workgroups reach more CUs (251 x 251 cells: 0.12 ms on 8 wavefronts, 0.18 on 16).
    if (DT > 0 && wmax >= WAVES_TOP && ntiles >= (int64_t)4 * ctx->num_cu * WAVES_TOP) {
        if (alds && fits(WAVES_TOP, true)) {
            waves = WAVES_TOP;
        } else if (!(alds && waves == WAVES_MAX) && fits(WAVES_TOP, false)) {
            waves = WAVES_TOP;
            alds = false;
        }
    }
    const size_t scratch = scratch_of(waves);
    const int head_doubles = (int)(small + (alds ? tri : 0));
    const size_t lds = sizeof(double) * ((size_t)head_doubles + scratch);
    if (lds > cap)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "k_gp_small: %zu bytes of LDS needed", lds);
    int64_t blocks = (ntiles + waves - 1) / waves;
    if (blocks > ctx->num_cu) blocks = ctx->num_cu;
    if (blocks < 1) blocks = 1;
    *nblocks = (int)blocks;
    SlAux aux{ctx->d_tri, ctx->d_net};
    unsigned long long* ticket = nullptr;
    if (ntiles > 4 * blocks * waves) {               // enough tiles per wavefront to balance
        ticket = ctx->d_ticket;
        SL_HIP_CHECK(ctx, hipMemsetAsync(ticket, 0, sizeof(unsigned long long), ctx->stream));
    }
    bool other_kernels = false;
    for (int k = 0; k < ctx->h_gp.nheads; ++k) other_kernels = other_kernels || ctx->gp_heads[k].d_kernel;
#define SL_GPS_GO(ALDS_, W_)                                                                       \
    do {                                                                                           \
        auto kern = other_kernels ? k_gp_small<GENERAL, DT, MT, ALDS_, W_, true>                   \
                                  : k_gp_small<GENERAL, DT, MT, ALDS_, W_, false>;                 \
        SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                 \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * W_), lds, ctx->stream, model,   \
                           ctx->h_gp, aux, lo, hi, ntiles, d_init_bits, d_values, d_neg_bits,      \
                           ctx->d_partials, d_dbg, head_doubles, d_points, ticket,              \
                           sl_diag_flags("SL_GPS_FLAGS"));                                      \
    } while (0)
    if constexpr (DT > 0) {
        if (waves == WAVES_MAX) { if (alds) SL_GPS_GO(true, WAVES_MAX); else SL_GPS_GO(false, WAVES_MAX); }
    }
    if constexpr (DT > 0) {
        if (waves == WAVES_TOP) { if (alds) SL_GPS_GO(true, WAVES_TOP); else SL_GPS_GO(false, WAVES_TOP); }
    }
    if constexpr (DT > 0) {
        if (waves == WAVES_TINY) { if (alds) SL_GPS_GO(true, WAVES_TINY); else SL_GPS_GO(false, WAVES_TINY); }
    }
    if (waves == WAVES_MIN) { if (alds) SL_GPS_GO(true, WAVES_MIN); else SL_GPS_GO(false, WAVES_MIN); }
#undef SL_GPS_GO
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_gp_small<general=%d, d=%d, m=%d, Linv in %s, %d wavefronts> (%d head(s))",
                   (int)GENERAL, DT, MT, alds ? "LDS" : "L2", waves, ctx->h_gp.nheads);
    return SL_OK;
}

int sl_gp_small_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                       const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits,
                       int* nblocks, double* d_dbg, const double* d_points) {
    const bool general = sl_model_is_general(model);
    const int variant = sl_dim_variant_of(model);
#define SL_GPS(G, D_, M_)                                                                          \
    return launch_small<G, D_, M_>(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks, \
                                   d_dbg, d_points)
    if (general) {
        if (variant == 2) SL_GPS(true, 2, 1);
        SL_GPS(true, 0, 0);
    }
    switch (variant) {
        case 1: SL_GPS(false, 1, 1);
        case 2: SL_GPS(false, 2, 1);
        case 3: SL_GPS(false, 3, 1);
        case 4: SL_GPS(false, 4, 1);
        default: SL_GPS(false, 0, 0);
    }
#undef SL_GPS
}
