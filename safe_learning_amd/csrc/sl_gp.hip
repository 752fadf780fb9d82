// sl_gp.hip - the GP-dynamics Lyapunov sweep for gfx950 (CDNA4): one fused kernel per sweep.
//
// Per cell the reference evaluates (safe_learning/functions.py:438-456, 507-515)
//     k_x   = variance * exp(-1/2 |(X - x*) / l|^2)            n values
//     a     = L^-1 k_x                                        n^2 flops  <- 96 % of the work
//     mean  = a^T alpha + m(x*),   var = k(x*,x*) - |a|^2,   err = beta sqrt(var)
// and then the decrease check of lyapunov.py:436-441.  Here:
//
//  * a = Linv k_x is a lower-triangular GEMM  [n x n] . [n x cells]  on the FP64 matrix cores
//    (v_mfma_f64_16x16x4_f64, 64-lane wavefronts).  Linv is pre-packed on the host into MFMA
//    A-fragment order, so every A operand is one coalesced 1 KiB load per wavefront, streamed
//    from L2 / Infinity Cache (the triangle is 4 MiB at n = 1024).
//  * A workgroup of W wavefronts owns a tile of C = 16*CB consecutive grid cells and a panel of
//    16*R*W rows; each wavefront keeps R x CB accumulator tiles (the whole a-panel) in
//    registers, k_x is produced cooperatively in 64-row chunks straight into B-fragment order
//    in LDS (double buffered, one barrier per chunk), so k_x never touches HBM.
//  * only |a|^2 per cell is needed, so the accumulators are squared and reduced in registers
//    (row order inside a tile is irrelevant); the mean uses alpha' = Linv^T alpha
//    (mean = k_x . alpha' + m(x*)), accumulated while k_x is generated.
//  * training inputs (pre-divided by the lengthscales) are staged once per workgroup in LDS.
//  * the decrease check, the 64-lane ballot for the mask and the (V, index) lexmin of the
//    failing cells run in the same kernel; HBM traffic is 8 B (V) + 1 bit per cell.
#include "sl_common.h"

typedef double sl_d4 __attribute__((ext_vector_type(4)));
typedef double sl_d2 __attribute__((ext_vector_type(2)));

#define SL_GP_SLABS_PER_CHUNK 16          // 64 training points per chunk
#ifndef SL_GP_STAGGER
#define SL_GP_STAGGER 1
#endif
#ifndef SL_GP_GEN_UNROLL
#define SL_GP_GEN_UNROLL 1             // more than one exp chain in flight only adds spills
#endif
#define SL_GP_DOUT_MAX SL_MAX_STATE_DIM

// configurations: {W wavefronts, R row blocks per wavefront, CB cell blocks}
//   cfg 0: small n (tests)   W=4 R=1 CB=1   panel   64 rows, 16 cells
//   cfg 1:                    W=8 R=8 CB=2   panel 1024 rows, 32 cells
//   cfg 2:                    W=8 R=4 CB=4   panel  512 rows, 64 cells; fast-path models run
//                             k_gp_sweep4 instead (sl_gp4.hip: 4x4x4 MFMAs, W=4 R=8 CB=4)
//   cfg 3:                    as cfg 2, always on this file's kernel (SL_GP_CFG=3: comparisons)
static const int kCfgW[4] = {4, 8, 8, 8};
static const int kCfgR[4] = {1, 8, 4, 4};
static const int kCfgCB[4] = {1, 2, 4, 4};

static inline int cfg_panel_rows(int cfg) { return 16 * kCfgR[cfg] * kCfgW[cfg]; }

// XSG: the scaled training inputs do not fit LDS next to the k_x buffers and are read from L2
// during generation (instantiated for the 64-cell-tile configuration only).
template <int W, int R, int CB, bool GENERAL, int DT, int MT, bool XSG = false>
__global__ __launch_bounds__(W * 64) void k_gp_sweep(
    const SlDevModel M, const SlGpDev gp, SlAux aux_arg, int64_t lo, int64_t hi, int64_t ntiles,
    const uint64_t* __restrict__ init_bits, const double* __restrict__ values,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    int xs_doubles, int alpha_doubles, const double* __restrict__ points, int nb_arg) {
    __shared__ SlTriLds<GENERAL> tri_lds;
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    constexpr int C = 16 * CB;
    constexpr int RP = 16 * R * W;                 // rows per panel
    constexpr int RB = R * W;                      // row blocks per panel
    constexpr int FRAGS = SL_GP_SLABS_PER_CHUNK * CB / W;   // k_x fragments per wavefront per chunk
    constexpr int KXBUF = SL_GP_SLABS_PER_CHUNK * CB * 64;  // doubles per k_x buffer
    static_assert(W % CB == 0, "W must be a multiple of CB");
    static_assert((SL_GP_SLABS_PER_CHUNK * CB) % W == 0, "fragments must divide evenly");

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xs_l = smem;                           // [p][n_pad]
    double* alpha_l = xs_l + xs_doubles;           // [n_pad][dout] when it fits (alpha_doubles > 0)
    double* kx_l = alpha_l + alpha_doubles;        // [2][KXBUF]
    double* part_ss = kx_l + 2 * KXBUF;            // [W][C]
    double* part_m = part_ss + W * C;              // [W][16][DOUT_MAX]
    // The table flavours check NB tiles at a time with one cell per lane: a table-V / table-
    // policy check costs thousands of cycles, and with one tile per check only C of the W * 64
    // lanes would work (measured 23 ms against 6.5 ms for the same sweep with a quadratic V).
    const int NB = GENERAL ? nb_arg : 1;
    double* cell_mean = part_m + W * 16 * SL_GP_DOUT_MAX;    // [NB][C][SL_D]
    double* cell_err = cell_mean + NB * C * SL_D;            // [NB][C][SL_D]
    uint64_t* sv = reinterpret_cast<uint64_t*>(cell_err + NB * C * SL_D);   // [W]
    int64_t* si = reinterpret_cast<int64_t*>(sv + W);                  // [W]
    int64_t* slot_tile = si + W;                                       // [NB] tiles of the batch

    const SlDims nd = sl_dims<DT, MT>(M);
    const int d = nd.d, p = nd.p;
    constexpr int DOUT_UNROLL = DT > 0 ? DT : SL_GP_DOUT_MAX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform
    const int gcb = wave % CB;                     // cell block this wavefront generates k_x for
    const int lcol = lane & 15, lk = lane >> 4;
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    int staged_head = -1;
    int nslots = 0;                                // tiles waiting for their check

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_base = lo + tile * C;
        const bool last_tile = tile + gridDim.x >= ntiles;
        if (tile_base >= hi) {                     // padding tile: only clears mask bits
            if (tid == 0) {
                if (C == 64) neg_bits[(tile_base - lo) >> 6] = 0ull;
                else if (C == 32) reinterpret_cast<uint32_t*>(neg_bits)[(tile_base - lo) >> 5] = 0u;
                else reinterpret_cast<uint16_t*>(neg_bits)[(tile_base - lo) >> 4] = 0;
            }
            if (!(last_tile && nslots > 0)) continue;
        } else {
        double* slot_mean = cell_mean + nslots * C * SL_D;
        double* slot_err = cell_err + nslots * C * SL_D;

        for (int h = 0; h < gp.nheads; ++h) {
            const SlGpHeadDev& hd = gp.head[h];
            const int n_pad = hd.n_pad, dout = hd.dout;
            const int nslab2 = hd.nslab2;
            const double variance = hd.variance;
            const sl_gp_kernel* __restrict__ kern = hd.kernel;    // null: the RBF of sl_gp_set_head
            const double* __restrict__ alphap = alpha_doubles > 0 ? alpha_l : hd.alpha;
            const double* __restrict__ mpack = hd.mpack;
            const double* __restrict__ xs_glob = hd.xs;
            if (staged_head != h) {
                __syncthreads();
                if (!XSG)
                    for (int k = tid; k < p * n_pad; k += W * 64) xs_l[k] = hd.xs[k];
                if (alpha_doubles > 0)
                    for (int k = tid; k < n_pad * dout; k += W * 64) alpha_l[k] = hd.alpha[k];
                staged_head = h;
                __syncthreads();
            }
            // GP input [x, policy(x)] / lengthscales of the cell this lane generates k_x for
            double xg[SL_P];
            {
                int64_t gidx = tile_base + 16 * gcb + lcol;
                gidx = gidx < hi ? gidx : hi - 1;
                double u[SL_M];
                sl_cell_state(M, d, gidx, points, xg);
                sl_policy_any<GENERAL>(M, nd, aux.tri, gidx, xg, u);
                sl_append_action(nd, u, xg);
#pragma unroll
                for (int q = 0; q < SL_P; ++q) xg[q] = (q < p) ? xg[q] * hd.inv_ls[q] : 0.0;
            }

            double ss[CB];
            double gmean[DOUT_UNROLL];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) ss[cb] = 0.0;
#pragma unroll
            for (int dd = 0; dd < DOUT_UNROLL; ++dd) gmean[dd] = 0.0;

            const int npanels = (hd.n + RP - 1) / RP;     // panels that hold training points
            for (int pan = 0; pan < npanels; ++pan) {
                sl_d4 acc[R][CB];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) acc[r][cb] = (sl_d4){0.0, 0.0, 0.0, 0.0};

                const int nchunks = (pan + 1) * (RP / 64);
                const int first_new_chunk = pan * (RP / 64);     // chunks not generated before

                // k_x chunk `ch` -> LDS buffer `buf`, B-fragment order [slab pair][cb][lane][2]
                auto generate = [&](int ch, int buf) {
                    const bool add_mean = ch >= first_new_chunk;
#pragma unroll SL_GP_GEN_UNROLL
                    for (int k = 0; k < FRAGS; ++k) {
                        const int f = wave + k * W;
                        const int s = f / CB;
                        const int j = 64 * ch + 4 * s + lk;
                        double z = 0.0;
                        double xa[SL_P];
#pragma unroll
                        for (int q = 0; q < SL_P; ++q) {
                            xa[q] = 0.0;
                            if (q < p) {
                                const double xv = XSG ? xs_glob[q * n_pad + j] : xs_l[q * n_pad + j];
                                const double dlt = xv - xg[q];
                                z = fma(dlt, dlt, z);
                                xa[q] = xv;
                            }
                        }
                        // (sum-of-products kernels: inputs unscaled, sl_gp_set_head_kernel)
                        const double kx = kern ? sl_kernel_eval(*kern, p, xa, xg) : variance * exp(-0.5 * z);
                        if (add_mean) {
#pragma unroll
                            for (int dd = 0; dd < DOUT_UNROLL; ++dd)
                                if (dd < dout) gmean[dd] = fma(kx, alphap[j * dout + dd], gmean[dd]);
                        }
                        kx_l[buf * KXBUF + ((((s >> 1) * CB + gcb) * 64 + lane) << 1) + (s & 1)] = kx;
                    }
                };

                // A-fragment address: uniform (row block, slab pair) offset + per-lane 16 bytes
                auto load_a = [&](int I, int s2abs) -> sl_d2 {
                    const double* base = mpack + ((size_t)I * nslab2 + (size_t)s2abs) * 128;
                    return *reinterpret_cast<const sl_d2*>(base + lane * 2);
                };

                generate(0, 0);
                __syncthreads();
                for (int ch = 0; ch < nchunks; ++ch) {
                    const int buf = ch & 1;
                    // slab pairs of this chunk on or below the diagonal, per owned row block;
                    // counts are even and non-decreasing in r, inactive row blocks come first
                    int cnt[R], rowblk[R];
                    int r0 = R;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int wsel = (r & 1) ? (W - 1 - wave) : wave;    // balance the triangle
                        rowblk[r] = pan * RB + r * W + wsel;
                        int c = 2 * rowblk[r] + 2 - 8 * ch;
                        c = c > 8 ? 8 : c;
                        cnt[r] = c > 0 ? c : 0;
                        if (cnt[r] > 0 && r0 == R) r0 = r;
                    }
                    // two-deep register queue of A fragments, primed before k_x generation so
                    // that the first loads fly while the next chunk's exponentials are computed
                    sl_d2 q0 = {0.0, 0.0}, q1 = {0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (r == r0) {
                            q0 = load_a(rowblk[r], 8 * ch);
                            q1 = load_a(rowblk[r], 8 * ch + 1);
                        }
                    }
                    // The two wavefronts that share a SIMD (w and w + 4) generate the next chunk at
                    // opposite ends of this chunk's MFMA phase, so that one of them always feeds
                    // the matrix pipe while the other computes exponentials on the VALU.
                    const bool gen_first = (W < 8) || ((wave & 4) == 0) || !SL_GP_STAGGER;
                    if (gen_first && ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                    const double* kxb = kx_l + buf * KXBUF;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int c = cnt[r];
                        for (int s2 = 0; s2 < c; ++s2) {
                            sl_d2 q2 = q1;
                            const int t = s2 + 2;
                            if (t < c) {
                                q2 = load_a(rowblk[r], 8 * ch + t);
                            } else if (r + 1 < R) {
                                q2 = load_a(rowblk[r + 1 < R ? r + 1 : r], 8 * ch + (t - c));
                            }
                            sl_d2 b2[CB];
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb)
                                b2[cb] = *reinterpret_cast<const sl_d2*>(
                                    kxb + (((s2 * CB + cb) * 64 + lane) << 1));
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb)
                                acc[r][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                    q0.x, b2[cb].x, acc[r][cb], 0, 0, 0);
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb)
                                acc[r][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                    q0.y, b2[cb].y, acc[r][cb], 0, 0, 0);
                            q0 = q1;
                            q1 = q2;
                        }
                    }
                    if (!gen_first && ch + 1 < nchunks) generate(ch + 1, buf ^ 1);
                    __syncthreads();
                }
                // |a|^2 of this panel's rows (row order inside a tile does not matter)
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        const sl_d4 t = acc[r][cb];
                        ss[cb] = fma(t.x, t.x, ss[cb]);
                        ss[cb] = fma(t.y, t.y, ss[cb]);
                        ss[cb] = fma(t.z, t.z, ss[cb]);
                        ss[cb] = fma(t.w, t.w, ss[cb]);
                    }
            }
            // ---- reduce over the 4 lane groups, then over wavefronts through LDS ---------------
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                ss[cb] += __shfl_xor(ss[cb], 16, 64);
                ss[cb] += __shfl_xor(ss[cb], 32, 64);
            }
#pragma unroll
            for (int dd = 0; dd < DOUT_UNROLL; ++dd) {
                gmean[dd] += __shfl_xor(gmean[dd], 16, 64);
                gmean[dd] += __shfl_xor(gmean[dd], 32, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) part_ss[wave * C + cb * 16 + lane] = ss[cb];
#pragma unroll
                for (int dd = 0; dd < DOUT_UNROLL; ++dd)
                    part_m[(wave * 16 + lane) * SL_GP_DOUT_MAX + dd] = gmean[dd];
            }
            __syncthreads();
            // (written behind this tile's barriers: every wavefront has left the previous check)
            if (tid == 0) slot_tile[nslots] = tile;
            if (tid < C) {
                double sumsq = 0.0;
                for (int w = 0; w < W; ++w) sumsq += part_ss[w * C + tid];
                double prior_var = variance;                               // functions.py:450
                if (kern) {                        // Kdiag of the cell's own input
                    double xc[SL_P], uc[SL_M];
                    int64_t gi = tile_base + tid;
                    gi = gi < hi ? gi : hi - 1;
                    sl_cell_state(M, d, gi, points, xc);
                    sl_policy_any<GENERAL>(M, nd, aux.tri, gi, xc, uc);
                    sl_append_action(nd, uc, xc);
                    prior_var = sl_kernel_diag(*kern, p, xc);
                }
                const double var = prior_var - sumsq;                      // functions.py:451
                const double e = gp.beta * sqrt(var);                      // functions.py:514
                const int cb = tid >> 4, cc = tid & 15;
                for (int dd = 0; dd < dout; ++dd) {
                    double mu = 0.0;
                    for (int w = cb; w < W; w += CB) mu += part_m[(w * 16 + cc) * SL_GP_DOUT_MAX + dd];
                    slot_mean[tid * SL_D + hd.col0 + dd] = mu;
                    slot_err[tid * SL_D + hd.col0 + dd] = e;
                }
            }
            __syncthreads();
        }

        ++nslots;
        }
        if (nslots < NB && !last_tile) continue;

        // ---- per-cell decrease check, mask word, failing-cell key -------------------------------
        // lane t of the workgroup: cell t % C of the batch's tile t / C
        const int slot = tid / C, cell = tid - slot * C;
        const bool in_batch = slot < nslots;
        const int64_t cbase = lo + slot_tile[in_batch ? slot : 0] * C;
        const int64_t idx = cbase + cell;
        const bool valid = in_batch && (idx < hi);
        bool negative = false;
        double v_x = 0.0;
        if (valid) {
            double x[SL_P], u[SL_M], prior[SL_D], mean[SL_D], err[SL_D];
            sl_cell_state(M, d, idx, points, x);
            sl_policy_any<GENERAL>(M, nd, aux.tri, idx, x, u);
            sl_append_action(nd, u, x);
            sl_rows_dot<SL_D, SL_P>(M.m.dynamics.matrix, d, p, x, prior);   // m(x*), functions.py:439
#pragma unroll
            for (int k = 0; k < SL_D; ++k) {
                if (k < d) {
                    mean[k] = cell_mean[tid * SL_D + k] + prior[k];
                    err[k] = cell_err[tid * SL_D + k];
                }
            }
            SlCellCheck c = sl_cell_check<SlSweepFlavour<GENERAL>::value>(M, d, aux, x, mean, err);
            negative = c.negative;
            v_x = values ? values[idx - lo] : c.v_x;       // ordering key: lyapunov.py:512
            if (dbg) {
                double* o = dbg + (idx - lo) * (2 + 2 * d);
                o[0] = c.decrease; o[1] = c.threshold;
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d) { o[2 + k] = mean[k]; o[2 + d + k] = err[k]; }
            }
        }
        if (wave * 64 < nslots * C) {              // wavefronts that hold cells of the batch
            const uint64_t word = __ballot(negative);
            bool init = false;
            if (init_bits && valid) init = (init_bits[(idx - lo) >> 6] >> ((idx - lo) & 63)) & 1ull;
            if (cell == 0 && in_batch) {               // first lane of a tile's piece of the word
                if (C == 64) neg_bits[(cbase - lo) >> 6] = word;
                else if (C == 32) reinterpret_cast<uint32_t*>(neg_bits)[(cbase - lo) >> 5] = (uint32_t)(word >> (lane & 32));
                else reinterpret_cast<uint16_t*>(neg_bits)[(cbase - lo) >> 4] = (uint16_t)(word >> (lane & 48));
            }
            const bool ok = negative || init;
            if (valid && !ok) sl_key_min(best_v, best_i, sl_vbits(v_x), idx);
        }
        nslots = 0;
        // (the batch buffers are rewritten only after the next tile's barriers)
    }
    __syncthreads();
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (tid == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// =============================================================================================
// host side
// =============================================================================================
static int choose_cfg(const sl_ctx* ctx, int n) {
    if (ctx->env.gp_cfg >= 0) return ctx->env.gp_cfg;
    if (n <= 256) return 0;
    return 2;        // 64-cell tiles: half the Linv traffic per MFMA of cfg 1, measured fastest
}

// h_kernel: a sum-of-products kernel (sl_gp_set_head_kernel; variance / lengthscales unused), or
// null for the RBF head
static int gp_set_head(sl_ctx* ctx, int head, int n, int p, int dout, int col0, const double* h_X,
                       const double* h_Linv, const double* h_alpha, double variance,
                       const double* h_lengthscales, const sl_gp_kernel* h_kernel) {
    if (!ctx || !h_X || !h_Linv || !h_alpha || !h_lengthscales)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head: NULL argument");
    if (head < 0 || head >= SL_MAX_GP_HEADS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head: head %d outside [0,%d)", head,
                       SL_MAX_GP_HEADS);
    if (n < 1 || p < 1 || p > SL_MAX_INPUT_DIM || dout < 1 || col0 < 0 ||
        col0 + dout > SL_MAX_STATE_DIM)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head: bad sizes n=%d p=%d dout=%d col0=%d",
                       n, p, dout, col0);
    if (!(variance > 0.0)) return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head: variance <= 0");
    for (int q = 0; q < p; ++q)
        if (!(h_lengthscales[q] > 0.0))
            return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head: lengthscale <= 0");
    const int cfg = choose_cfg(ctx, n);
    const int rp = cfg_panel_rows(cfg);
    const int n_pad = ((n + rp - 1) / rp) * rp;
    const int nslab2 = n_pad / 8;

    std::vector<double> xs((size_t)p * n_pad, 0.0), alphap((size_t)n_pad * dout, 0.0);
    std::vector<double> mpack((size_t)n_pad * n_pad, 0.0);
    for (int j = 0; j < n; ++j)
        for (int q = 0; q < p; ++q) xs[(size_t)q * n_pad + j] = h_X[(size_t)j * p + q] / h_lengthscales[q];
    // alpha' = Linv^T alpha  (so that mean = k_x . alpha' = a . alpha, functions.py:442)
    for (int i = 0; i < n; ++i) {
        for (int dd = 0; dd < dout; ++dd) {
            const double ai = h_alpha[(size_t)i * dout + dd];
            if (ai == 0.0) continue;
            for (int j = 0; j <= i; ++j) alphap[(size_t)j * dout + dd] += h_Linv[(size_t)i * n + j] * ai;
        }
    }
    // MFMA A fragments: [row block I][slab pair S2][lane][2]; A[i = lane & 15][k = lane >> 4]
    for (int I = 0; I < n_pad / 16; ++I) {
        for (int S2 = 0; S2 < nslab2; ++S2) {
            if (8 * S2 > 16 * I + 15) continue;                      // strictly above the diagonal
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * I + (l & 15);
                for (int e = 0; e < 2; ++e) {
                    const int col = 4 * (2 * S2 + e) + (l >> 4);
                    double v = 0.0;
                    if (row < n && col <= row) v = h_Linv[(size_t)row * n + col];
                    mpack[(((size_t)I * nslab2 + S2) * 64 + l) * 2 + e] = v;
                }
            }
        }
    }
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ++ctx->dynamics_token;                         // successor cache: the posterior mean changes
    SlGpHeadHost& hh = ctx->gp_heads[head];
    if (hh.d_xs) (void)hipFree(hh.d_xs);
    if (hh.d_mpack) (void)hipFree(hh.d_mpack);
    if (hh.d_alpha) (void)hipFree(hh.d_alpha);
    if (hh.d_kernel) (void)hipFree(hh.d_kernel);
    hh.d_xs = hh.d_mpack = hh.d_alpha = nullptr;
    hh.d_kernel = nullptr;
    if (h_kernel) {
        SL_HIP_CHECK(ctx, hipMalloc(&hh.d_kernel, sizeof(sl_gp_kernel)));
        SL_HIP_CHECK(ctx, hipMemcpy(hh.d_kernel, h_kernel, sizeof(sl_gp_kernel), hipMemcpyHostToDevice));
        hh.h_kernel = *h_kernel;
    }
    SL_HIP_CHECK(ctx, hipMalloc(&hh.d_xs, xs.size() * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMalloc(&hh.d_mpack, mpack.size() * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMalloc(&hh.d_alpha, alphap.size() * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMemcpy(hh.d_xs, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
    SL_HIP_CHECK(ctx, hipMemcpy(hh.d_mpack, mpack.data(), mpack.size() * sizeof(double), hipMemcpyHostToDevice));
    SL_HIP_CHECK(ctx, hipMemcpy(hh.d_alpha, alphap.data(), alphap.size() * sizeof(double), hipMemcpyHostToDevice));
    hh.set = true; hh.n = n; hh.n_pad = n_pad; hh.p = p; hh.dout = dout; hh.col0 = col0; hh.cfg = cfg;

    SlGpHeadDev& dv = ctx->h_gp.head[head];
    memset(&dv, 0, sizeof(dv));
    dv.n = n; dv.n_pad = n_pad; dv.p = p; dv.dout = dout; dv.col0 = col0; dv.nslab2 = nslab2;
    dv.variance = variance;
    for (int q = 0; q < p; ++q) dv.inv_ls[q] = 1.0 / h_lengthscales[q];
    for (int q = 0; q < p; ++q) ctx->gp_heads[head].lengthscales[q] = h_lengthscales[q];
    dv.xs = hh.d_xs; dv.mpack = hh.d_mpack; dv.alpha = hh.d_alpha;
    dv.kernel = hh.d_kernel;
    return SL_OK;
}

extern "C" int sl_gp_set_head(sl_ctx* ctx, int head, int n, int p, int dout, int col0,
                              const double* h_X, const double* h_Linv, const double* h_alpha,
                              double variance, const double* h_lengthscales) {
    return gp_set_head(ctx, head, n, p, dout, col0, h_X, h_Linv, h_alpha, variance, h_lengthscales,
                       nullptr);
}

extern "C" int sl_gp_set_head_kernel(sl_ctx* ctx, int head, int n, int p, int dout, int col0,
                                     const double* h_X, const double* h_Linv, const double* h_alpha,
                                     const sl_gp_kernel* h_kernel) {
    if (!h_kernel) return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: NULL kernel");
    if (h_kernel->nfactors < 1 || h_kernel->nfactors > SL_KERNEL_MAX_FACTORS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: %d factors outside [1,%d]",
                       h_kernel->nfactors, SL_KERNEL_MAX_FACTORS);
    int product = 0;
    for (int f = 0; f < h_kernel->nfactors; ++f) {
        const sl_gp_kernel_factor& fac = h_kernel->factor[f];
        if (fac.kind != SL_KERNEL_RBF && fac.kind != SL_KERNEL_MATERN32 && fac.kind != SL_KERNEL_LINEAR)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: factor %d has kind %d", f, fac.kind);
        if (fac.product != product && fac.product != product + 1)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: factor %d belongs to product "
                           "%d after product %d (the factors of a product are adjacent, products "
                           "numbered in order)", f, fac.product, product);
        if (f == 0 && fac.product != 0)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: the first product is number 0");
        product = fac.product;
        for (int q = 0; q < SL_MAX_INPUT_DIM; ++q)
            if (!(fac.variance[q] >= 0.0) || !(fac.inv_lengthscales[q] >= 0.0))
                return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_set_head_kernel: factor %d: negative or "
                               "NaN variance / inverse lengthscale", f);
    }
    double ones[SL_MAX_INPUT_DIM];
    for (int q = 0; q < SL_MAX_INPUT_DIM; ++q) ones[q] = 1.0;
    return gp_set_head(ctx, head, n, p, dout, col0, h_X, h_Linv, h_alpha, 1.0, ones, h_kernel);
}

// row n of the extended L^-1 into the A-fragment layout, the rank-one term of alpha', the new
// scaled input column
__global__ void k_gp_append(const double* __restrict__ row, int n, int n_pad, int nslab2, int p,
                            int dout, const double* __restrict__ x_scaled,
                            const double* __restrict__ alpha_new, double* __restrict__ mpack,
                            double* __restrict__ alphap, double* __restrict__ xs) {
    const int I = n >> 4, r16 = n & 15;
    const int count = n + 1 > p ? n + 1 : p;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < count; c += gridDim.x * blockDim.x) {
        if (c <= n) {
            const double v = row[c];
            const int S2 = c >> 3, e = (c >> 2) & 1, k = c & 3;
            mpack[(((size_t)I * nslab2 + S2) * 64 + 16 * k + r16) * 2 + e] = v;
            for (int dd = 0; dd < dout; ++dd) alphap[(size_t)c * dout + dd] += v * alpha_new[dd];
        }
        if (c < p) xs[(size_t)c * n_pad + n] = x_scaled[c];
    }
}

extern "C" int sl_gp_append_point(sl_ctx* ctx, int head, const double* h_x, const double* h_linv_row,
                                  const double* h_alpha_new) {
    if (!ctx || !h_x || !h_linv_row || !h_alpha_new)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_append_point: NULL argument");
    if (head < 0 || head >= SL_MAX_GP_HEADS || !ctx->gp_heads[head].set)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_append_point: head %d not set", head);
    SlGpHeadHost& hh = ctx->gp_heads[head];
    SlGpHeadDev& dv = ctx->h_gp.head[head];
    const int n = hh.n;
    if (n + 1 > hh.n_pad)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "sl_gp_append_point: capacity %d of head %d is "
                                                "exhausted (upload the head again)", hh.n_pad, head);
    ++ctx->dynamics_token;                         // successor cache: the posterior mean changes
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t need = sizeof(double) * (size_t)(n + 1 + hh.p + hh.dout);
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, need));
        ctx->scratch_bytes = need;
    }
    std::vector<double> stage((size_t)(n + 1 + hh.p + hh.dout));
    for (int c = 0; c <= n; ++c) stage[c] = h_linv_row[c];
    for (int q = 0; q < hh.p; ++q) stage[n + 1 + q] = h_x[q] / hh.lengthscales[q];   // as sl_gp_set_head
    for (int dd = 0; dd < hh.dout; ++dd) stage[n + 1 + hh.p + dd] = h_alpha_new[dd];
    double* d_stage = reinterpret_cast<double*>(ctx->d_scratch);
    SL_HIP_CHECK(ctx, hipMemcpy(d_stage, stage.data(), need, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gp_append, dim3((n + hh.p + 256) / 256), dim3(256), 0, ctx->stream, d_stage, n,
                       hh.n_pad, dv.nslab2, hh.p, hh.dout, d_stage + n + 1, d_stage + n + 1 + hh.p,
                       hh.d_mpack, hh.d_alpha, hh.d_xs);
    SL_HIP_CHECK(ctx, hipGetLastError());
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hh.n = n + 1;
    dv.n = n + 1;
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_gp, &ctx->h_gp, sizeof(SlGpDev), hipMemcpyHostToDevice));
    return SL_OK;
}

extern "C" int sl_debug_gp_inputs(sl_ctx* ctx, int head, double* h_xs) {
    if (!ctx || !h_xs) return sl_fail(ctx, SL_ERR_INVALID, "sl_debug_gp_inputs: NULL argument");
    if (head < 0 || head >= SL_MAX_GP_HEADS || !ctx->gp_heads[head].set)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_debug_gp_inputs: head %d not set", head);
    const SlGpHeadHost& hh = ctx->gp_heads[head];
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < hh.p; ++q)
        SL_HIP_CHECK(ctx, hipMemcpy(h_xs + (size_t)q * hh.n, hh.d_xs + (size_t)q * hh.n_pad,
                                    sizeof(double) * hh.n, hipMemcpyDeviceToHost));
    return SL_OK;
}

extern "C" int sl_gp_configure(sl_ctx* ctx, int nheads, double beta) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_gp_configure: NULL context");
    if (nheads < 1 || nheads > SL_MAX_GP_HEADS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_configure: %d heads", nheads);
    int cfg = -1;
    for (int h = 0; h < nheads; ++h) {
        if (!ctx->gp_heads[h].set)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_gp_configure: head %d not set", h);
        if (cfg < 0) cfg = ctx->gp_heads[h].cfg;
        if (cfg != ctx->gp_heads[h].cfg)
            return sl_fail(ctx, SL_ERR_UNSUPPORTED,
                           "sl_gp_configure: heads need the same kernel configuration "
                           "(training-set sizes too different)");
    }
    ctx->gp_cfg = cfg;
    if (ctx->h_gp.nheads != nheads) ++ctx->dynamics_token;
    ctx->h_gp.nheads = nheads;
    ctx->h_gp.beta = beta;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_gp, &ctx->h_gp, sizeof(SlGpDev), hipMemcpyHostToDevice));
    return SL_OK;
}

template <int W, int R, int CB, bool GENERAL, int DT, int MT, bool XSG = false>
static int launch_cfg(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                      const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits, int* nblocks, double* d_dbg,
                      const double* d_points) {
    constexpr int C = 16 * CB;
    const int64_t nwords = (hi - lo + 63) / 64;
    const int64_t ntiles = nwords * (64 / C);
    int xs_doubles = 0;
    const int p = model.in_dim;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) {
        if (ctx->gp_heads[h].p != p)
            return sl_fail(ctx, SL_ERR_INVALID, "GP head %d has input dim %d, model has %d", h,
                           ctx->gp_heads[h].p, p);
        const int v = p * ctx->gp_heads[h].n_pad;
        xs_doubles = v > xs_doubles ? v : xs_doubles;
    }
    xs_doubles = (xs_doubles + 1) & ~1;            // keep the k_x buffers 16-byte aligned
    if (XSG) xs_doubles = 0;                       // training inputs stay in global memory / L2
    constexpr size_t LDS_CAP = 160 * 1024 - (GENERAL ? sizeof(SlTriLds<true>) + 64 : 0);
    const auto lds_for = [&](int nb) {
        return sizeof(double) * ((size_t)xs_doubles + 2 * SL_GP_SLABS_PER_CHUNK * CB * 64 + W * C +
                                 W * 16 * SL_GP_DOUT_MAX + 2 * (size_t)nb * C * SL_D + 2 * W + nb);
    };
    // tiles per check of the table flavours: one cell per lane when LDS has the room
    int nb = GENERAL ? (W * 64) / C : 1;
    while (nb > 1 && lds_for(nb) > LDS_CAP) nb /= 2;
    size_t lds = lds_for(nb);
    // alpha' next to the training inputs when LDS has room (the mean accumulation of the k_x
    // generation then reads LDS instead of L2)
    int alpha_doubles = 0;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) {
        const int v = ctx->gp_heads[h].n_pad * ctx->gp_heads[h].dout;
        alpha_doubles = v > alpha_doubles ? v : alpha_doubles;
    }
    alpha_doubles = (alpha_doubles + 1) & ~1;
    // (the general flavours also hold both table descriptors in static LDS)
    if (lds + sizeof(double) * alpha_doubles <= LDS_CAP) lds += sizeof(double) * alpha_doubles;
    else alpha_doubles = 0;
    if (lds > LDS_CAP)
        return sl_fail(ctx, SL_ERR_UNSUPPORTED, "GP training set too large for LDS staging "
                                                "(%zu bytes needed)", lds);
    auto kern = k_gp_sweep<W, R, CB, GENERAL, DT, MT, XSG>;
    SL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = ntiles < ctx->num_cu ? ntiles : ctx->num_cu;
    if (blocks > SL_MAX_GRID) blocks = SL_MAX_GRID;
    *nblocks = (int)blocks;
    SlAux aux{ctx->d_tri, ctx->d_net};
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W * 64), lds, ctx->stream, model,
                       ctx->h_gp, aux, lo, hi, ntiles, d_init_bits, d_values, d_neg_bits, ctx->d_partials, d_dbg,
                       xs_doubles, alpha_doubles, d_points, nb);
    SL_HIP_CHECK(ctx, hipGetLastError());
    sl_note_kernel(ctx, false, "k_gp_sweep<W=%d, R=%d, CB=%d, general=%d, d=%d, m=%d, xs_global=%d>", W, R,
                   CB, (int)GENERAL, DT, MT, (int)XSG);
    return SL_OK;
}

int sl_gp_sweep_launch(sl_ctx* ctx, const SlDevModel& model, int64_t lo, int64_t hi,
                       const uint64_t* d_init_bits, const double* d_values, uint64_t* d_neg_bits, int* nblocks, double* d_dbg,
                       const double* d_points) {
    if (ctx->h_gp.nheads < 1)
        return sl_fail(ctx, SL_ERR_INVALID, "GP dynamics selected but sl_gp_configure not called");
    int covered = 0;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) covered += ctx->gp_heads[h].dout;
    if (covered != model.m.grid.d)
        return sl_fail(ctx, SL_ERR_INVALID, "GP heads cover %d outputs, state dimension is %d",
                       covered, model.m.grid.d);
    const bool general = sl_model_is_general(model);
    const int variant = sl_dim_variant_of(model);
    // heads with a sum-of-products kernel (sl_gp_set_head_kernel): k_gp_small evaluates those
    bool other_kernels = false;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) other_kernels = other_kernels || ctx->gp_heads[h].d_kernel;
    // (k_gp_small and k_gp_sweep evaluate them; k_gp_sweep4 generates RBF values by recurrence)
    if (other_kernels && sl_gp_small_supports(ctx, model))
        return sl_gp_small_launch(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,
                                  d_dbg, d_points);
    // 225 .. 256 training points per head (capacity exactly one 256-row panel of k_gp_sweep4): the
    // factor no longer fits LDS, k_gp_small waits for its fragments from L2 (1.9 ms at 256 points on
    // 1024^2 cells, 0.6 ms at 128) while k_gp_sweep4 prefetches them two slab pairs ahead - fast-path
    // models with RBF heads take it on ONE panel (SL_GP4_ONE_PANEL=0: keep k_gp_small).  k_gp_small
    // pays for the row blocks that hold training points, the panel for its 256 rows: at 224 points
    // the two meet, below k_gp_small wins (round 6, sixteen wavefronts and the short pass first:
    // 1024^2 cells 1.39 / 1.54 / 1.71 / 1.91 ms at 200 / 224 / 240 / 256 points against 1.50 - 1.52 on the
    // panel; 48^4: 6.90 / 7.78 / 9.66 against 7.95 - profiles/r06_one_panel_ab.txt; round 5 had the
    // crossing at 200, profiles/r05_one_panel_ab.txt).
    if (ctx->gp_cfg == 0 && !other_kernels && sl_gp4_supports(model)) {
        bool one_panel = true;
        for (int h = 0; h < ctx->h_gp.nheads; ++h)
            one_panel = one_panel && ctx->gp_heads[h].n_pad == 256 && ctx->gp_heads[h].n > 224;
        if (ctx->env.gp4_one_panel == 0) one_panel = false;
        if (one_panel)
            return sl_gp4_sweep_launch(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,
                                       d_dbg, d_points);
    }
    // small training sets (one head, capacity <= 256 points): a wavefront per 64-cell tile
    if (ctx->gp_cfg == 0 && sl_gp_small_supports(ctx, model))
        return sl_gp_small_launch(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,
                                  d_dbg, d_points);
    if (ctx->gp_cfg == 2 && !other_kernels && sl_gp4_supports(model))
        return sl_gp4_sweep_launch(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits, nblocks,
                                   d_dbg, d_points);
    // training inputs too large to sit in LDS beside the k_x buffers: 64-cell-tile configuration
    // with the inputs read from L2 (fast-path models with 2 or 4 state dimensions)
    {
        int xs_max = 0;
        for (int h = 0; h < ctx->h_gp.nheads; ++h) {
            const int v = model.in_dim * ctx->gp_heads[h].n_pad;
            xs_max = v > xs_max ? v : xs_max;
        }
        const size_t fixed = sizeof(double) * (2 * SL_GP_SLABS_PER_CHUNK * 4 * 64 + 8 * 64 +
                                               8 * 16 * SL_GP_DOUT_MAX + 2 * 64 * SL_D + 2 * 8);
        if (ctx->gp_cfg >= 2 && fixed + sizeof(double) * xs_max > 160 * 1024) {
            if (!general && variant == 4)
                return launch_cfg<8, 4, 4, false, 4, 1, true>(ctx, model, lo, hi, d_init_bits, d_values,
                                                              d_neg_bits, nblocks, d_dbg, d_points);
            if (!general && variant == 2)
                return launch_cfg<8, 4, 4, false, 2, 1, true>(ctx, model, lo, hi, d_init_bits, d_values,
                                                              d_neg_bits, nblocks, d_dbg, d_points);
            return launch_cfg<8, 4, 4, true, 0, 0, true>(ctx, model, lo, hi, d_init_bits, d_values,
                                                         d_neg_bits, nblocks, d_dbg, d_points);
        }
    }
#define SL_GP_LAUNCH(W_, R_, CB_, G, D_, M_)                                                      \
    return launch_cfg<W_, R_, CB_, G, D_, M_>(ctx, model, lo, hi, d_init_bits, d_values, d_neg_bits,     \
                                              nblocks, d_dbg, d_points)
#define SL_GP_CASE(id, W_, R_, CB_)                                                               \
    case id:                                                                                      \
        if (general) {                                                                            \
            if (variant == 2) { SL_GP_LAUNCH(W_, R_, CB_, true, 2, 1); }                          \
            SL_GP_LAUNCH(W_, R_, CB_, true, 0, 0);                                                \
        }                                                                                         \
        switch (variant) {                                                                        \
            case 1: SL_GP_LAUNCH(W_, R_, CB_, false, 1, 1);                                       \
            case 2: SL_GP_LAUNCH(W_, R_, CB_, false, 2, 1);                                       \
            case 3: SL_GP_LAUNCH(W_, R_, CB_, false, 3, 1);                                       \
            case 4: SL_GP_LAUNCH(W_, R_, CB_, false, 4, 1);                                       \
            default: SL_GP_LAUNCH(W_, R_, CB_, false, 0, 0);                                      \
        }
    switch (ctx->gp_cfg) {
        SL_GP_CASE(0, 4, 1, 1)
        SL_GP_CASE(1, 8, 8, 2)
        SL_GP_CASE(2, 8, 4, 4)
        SL_GP_CASE(3, 8, 4, 4)
    }
#undef SL_GP_CASE
#undef SL_GP_LAUNCH
    return sl_fail(ctx, SL_ERR_INVALID, "bad GP kernel configuration %d", ctx->gp_cfg);
}

// =============================================================================================
// diagnostics
// =============================================================================================
__global__ void k_debug_mfma(const double* __restrict__ a, const double* __restrict__ b,
                             double* __restrict__ dout) {
    const int l = threadIdx.x;
    sl_d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(l & 15) * 4 + (l >> 4)], b[(l >> 4) * 16 + (l & 15)],
                                               acc, 0, 0, 0);
    // assumed C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
    dout[((l >> 4) + 0) * 16 + (l & 15)] = acc.x;
    dout[((l >> 4) + 4) * 16 + (l & 15)] = acc.y;
    dout[((l >> 4) + 8) * 16 + (l & 15)] = acc.z;
    dout[((l >> 4) + 12) * 16 + (l & 15)] = acc.w;
}

extern "C" int sl_debug_mfma(sl_ctx* ctx, const double* h_a, const double* h_b, double* h_d) {
    if (!ctx || !h_a || !h_b || !h_d) return sl_fail(ctx, SL_ERR_INVALID, "sl_debug_mfma: NULL");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    double *da, *db, *dd;
    SL_HIP_CHECK(ctx, hipMalloc(&da, 64 * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMalloc(&db, 64 * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMalloc(&dd, 256 * sizeof(double)));
    SL_HIP_CHECK(ctx, hipMemcpy(da, h_a, 64 * sizeof(double), hipMemcpyHostToDevice));
    SL_HIP_CHECK(ctx, hipMemcpy(db, h_b, 64 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_mfma, dim3(1), dim3(64), 0, ctx->stream, da, db, dd);
    SL_HIP_CHECK(ctx, hipGetLastError());
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SL_HIP_CHECK(ctx, hipMemcpy(h_d, dd, 256 * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    return SL_OK;
}

// v_mfma_f64_4x4x4_4b_f64 with per-lane operands; mode 0: plain, 1..4: cbsz = 2, abid = mode - 1
__global__ void k_debug_mfma4(const double* a, const double* b, const double* c, int mode, double* d) {
    const int l = threadIdx.x + 64 * blockIdx.x;
    const double av = a[l], bv = b[l], cv = c[l];
    double r;
    switch (mode) {
    case 1: r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 0, 0); break;
    case 2: r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 1, 0); break;
    case 3: r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 2, 0); break;
    case 4: r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 3, 0); break;
    default: r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 0, 0, 0); break;
    }
    d[l] = r;
}

// nwaves independent problems of 64 lanes each
extern "C" int sl_debug_mfma4(sl_ctx* ctx, int nwaves, const double* h_a, const double* h_b,
                              const double* h_c, int mode, double* h_d) {
    if (!ctx || !h_a || !h_b || !h_c || !h_d || nwaves < 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_debug_mfma4: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * 64 * (size_t)nwaves;
    double* buf;
    SL_HIP_CHECK(ctx, hipMalloc(&buf, 4 * bytes));
    SL_HIP_CHECK(ctx, hipMemcpy(buf, h_a, bytes, hipMemcpyHostToDevice));
    SL_HIP_CHECK(ctx, hipMemcpy(buf + 64 * nwaves, h_b, bytes, hipMemcpyHostToDevice));
    SL_HIP_CHECK(ctx, hipMemcpy(buf + 128 * nwaves, h_c, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_mfma4, dim3(nwaves), dim3(64), 0, ctx->stream, buf, buf + 64 * nwaves,
                       buf + 128 * nwaves, mode, buf + 192 * nwaves);
    SL_HIP_CHECK(ctx, hipGetLastError());
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SL_HIP_CHECK(ctx, hipMemcpy(h_d, buf + 192 * nwaves, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(buf);
    return SL_OK;
}

// FP64 rate probes: which = 0 MFMA only, 1 VALU FMA only, 2 both in the same wavefront.
// Also reports the shader clock actually sustained (s_memtime ticks / s_memrealtime ticks).
template <int WHICH>
__global__ __launch_bounds__(256) void k_fp64_rate(int iters, double* sink, long long* clocks) {
    sl_d4 acc[8];
    __shared__ __attribute__((aligned(16))) double probe_lds[32 * 128];
    if (WHICH == 8) {
        for (int k = threadIdx.x; k < 32 * 128; k += blockDim.x) probe_lds[k] = 1.0 + k * 1e-9;
        __syncthreads();
    }
    double v[8], w16[16], w64[(WHICH == 6 || WHICH == 7 || WHICH == 8) ? 64 : 1];
#pragma unroll
    for (int k = 0; k < ((WHICH == 6 || WHICH == 7 || WHICH == 8) ? 64 : 1); ++k) w64[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] = (sl_d4){0.0, 0.0, 0.0, 0.0}; v[k] = threadIdx.x * 1e-3 + k; }
#pragma unroll
    for (int k = 0; k < 16; ++k) w16[k] = 0.0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (WHICH == 0 || WHICH == 2) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
            if (WHICH == 3)               // accumulators pinned to the AccVGPR half of the file
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
            if (WHICH == 4)               // 4 blocks of 4x4x4: 512 flops per instruction
                v[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v[k], 0, 0, 0);
            if (WHICH == 5) {             // 4x4x4 with four distinct A and B operands, 16 accumulators
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    w16[(2 * k + q) & 15] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                        v[(k + q) & 3], v[4 + ((k >> 1) & 3)], w16[(2 * k + q) & 15], 0, 0, 0);
            }
            if (WHICH == 8) {             // 4x4x4 MFMAs fed by LDS reads: one ds_read_b128 per 8 MFMAs
                const sl_d2 t = *reinterpret_cast<const sl_d2*>(&probe_lds[((it * 8 + k) & 31) * 128 + 2 * (threadIdx.x & 63)]);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    w64[8 * k + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(q & 1 ? t.x : t.y, w16[q + 8 * (k >> 2)],
                                                                          w64[8 * k + q], 0, 0, 0);
            }
            if (WHICH == 7) {             // 4x4x4 MFMAs with FP64 VALU work between them (same wavefront)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    w64[8 * k + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[k & 3], w16[q + 8 * (k >> 2)],
                                                                          w64[8 * k + q], 0, 0, 0);
                    acc[q].x = fma(acc[q].x, b, a);
                    acc[q].y = fma(acc[q].y, b, a);
                }
            }
            if (WHICH == 6) {             // the GP kernel's register shape: 64 accumulators, 4 A x 8 B operands
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    w64[8 * k + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[k & 3], w16[q + 8 * (k >> 2)],
                                                                          w64[8 * k + q], 0, 0, 0);
            }
            if (WHICH == 1 || WHICH == 2) {
#pragma unroll
                for (int rep = 0; rep < 16; ++rep) v[(k + rep) & 7] = fma(v[(k + rep) & 7], b, a);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w + v[k] + w16[k] + w16[k + 8];
#pragma unroll
    for (int k = 0; k < ((WHICH == 6 || WHICH == 7 || WHICH == 8) ? 64 : 1); ++k) s += w64[k];
    const long long c1 = clock64(), w1 = wall_clock64();
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

extern "C" int sl_debug_fp64_rate(sl_ctx* ctx, int which, int iters, double* h_out) {
    if (!ctx || !h_out || which < 0 || which > 8 || iters < 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_debug_fp64_rate: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int per_cu = (ctx->env.probe_blocks_per_cu >= 1 && ctx->env.probe_blocks_per_cu <= 8)
                           ? ctx->env.probe_blocks_per_cu : 2;
    double* sink;
    long long* clocks;
    SL_HIP_CHECK(ctx, hipMalloc(&sink, sizeof(double)));
    SL_HIP_CHECK(ctx, hipMalloc(&clocks, 2 * sizeof(long long)));
    hipEvent_t e0, e1;
    SL_HIP_CHECK(ctx, hipEventCreate(&e0));
    SL_HIP_CHECK(ctx, hipEventCreate(&e1));
    const int blocks = ctx->num_cu * per_cu;     // per_cu blocks of 4 wavefronts per CU
    for (int rep = 0; rep < 2; ++rep) {
        SL_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
        if (which == 0) hipLaunchKernelGGL(k_fp64_rate<0>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 1) hipLaunchKernelGGL(k_fp64_rate<1>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 2) hipLaunchKernelGGL(k_fp64_rate<2>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 3) hipLaunchKernelGGL(k_fp64_rate<3>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 4) hipLaunchKernelGGL(k_fp64_rate<4>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 5) hipLaunchKernelGGL(k_fp64_rate<5>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 6) hipLaunchKernelGGL(k_fp64_rate<6>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else if (which == 7) hipLaunchKernelGGL(k_fp64_rate<7>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        else hipLaunchKernelGGL(k_fp64_rate<8>, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink, clocks);
        SL_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
        SL_HIP_CHECK(ctx, hipEventSynchronize(e1));
    }
    float ms = 0.f;
    SL_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
    long long hc[2] = {0, 0};
    SL_HIP_CHECK(ctx, hipMemcpy(hc, clocks, sizeof(hc), hipMemcpyDeviceToHost));
    const double waves = (double)blocks * 4.0;
    double flops = 0.0;
    if (which == 0 || which == 2 || which == 3) flops += waves * (double)iters * 8.0 * (2.0 * 16 * 16 * 4);
    if (which == 4) flops += waves * (double)iters * 8.0 * (4 * 2.0 * 4 * 4 * 4);
    if (which == 5) flops += waves * (double)iters * 16.0 * (4 * 2.0 * 4 * 4 * 4);
    if (which == 6 || which == 7 || which == 8) flops += waves * (double)iters * 64.0 * (4 * 2.0 * 4 * 4 * 4);
    if (which == 7) flops += waves * (double)iters * 64.0 * 2.0 * 64.0 * 2.0;
    if (which == 1 || which == 2) flops += waves * (double)iters * 8.0 * 16.0 * 64.0 * 2.0;
    h_out[0] = flops / (ms * 1e-3) / 1e12;                       // TFLOP/s
    h_out[1] = hc[1] > 0 ? 100.0 * (double)hc[0] / (double)hc[1] : 0.0;   // shader MHz (100 MHz ref)
    h_out[2] = (double)hc[0] / ((double)iters * 8.0 * per_cu);   // shader cycles per MFMA (or per 16 FMA) slot per SIMD
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink); (void)hipFree(clocks);
    return SL_OK;
}
