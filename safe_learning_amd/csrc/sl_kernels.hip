// sl_kernels.hip - context, model upload and the streaming (HBM-bound) passes of the Lyapunov
// sweep for gfx950: values, deterministic-dynamics decrease check, safe-set finalisation,
// radix-select histogram, bit/byte mask conversion.  The GP (MFMA) sweep lives in sl_gp.hip,
// the dynamic-programming sweep in sl_bellman.hip.
//
// Launch geometry: 256-thread workgroups (4 wavefronts of 64), a grid capped at 2048 blocks
// (= 8 per CU) that walks the cell range in 256-cell strides, so consecutive lanes own
// consecutive flat indices: value stores are 512 B per wavefront instruction, each wavefront
// produces exactly one 64-bit mask word with one ballot.
#include <stdarg.h>
#include <stdlib.h>

#include "sl_common.h"

thread_local std::string g_sl_last_error;

void sl_note_kernel(sl_ctx* ctx, bool append, const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (append && ctx->last_kernel[0]) {
        const size_t used = strlen(ctx->last_kernel);
        snprintf(ctx->last_kernel + used, sizeof(ctx->last_kernel) - used, " + %s", buf);
    } else {
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "%s", buf);
    }
}

extern "C" const char* sl_last_kernel(const sl_ctx* ctx) { return ctx ? ctx->last_kernel : ""; }

int sl_fail(sl_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_sl_last_error = buf;
    if (ctx) ctx->error = buf;
    return code;
}

// =============================================================================================
// context
// =============================================================================================
extern "C" int sl_version(void) { return 100; }

static int env_int(const char* name) {
    const char* v = getenv(name);
    if (!v || !v[0]) return -1;
    const int i = atoi(v);
    return i < 0 ? 0 : i;
}

void sl_env_read(SlEnv* e) {
    e->gp_cfg = env_int("SL_GP_CFG");
    if (e->gp_cfg > 3) e->gp_cfg = -1;
    e->gp_small = env_int("SL_GP_SMALL");
    e->gp_small_waves = env_int("SL_GP_SMALL_WAVES");
    e->gp_small_split = env_int("SL_GP_SMALL_SPLIT");
    e->det_rows = env_int("SL_DET_ROWS");
    e->gp4_one_panel = env_int("SL_GP4_ONE_PANEL");
    e->gp4_seeds = env_int("SL_GP4_SEEDS");
    e->gp4_tickets = env_int("SL_GP4_TICKETS");
    e->bellman_mfma = env_int("SL_BELLMAN_MFMA");
    e->bellman4 = env_int("SL_BELLMAN4");
    e->bellman4_policy = env_int("SL_BELLMAN4_POLICY");
    e->bellman4_policy_cache = env_int("SL_BELLMAN4_POLICY_CACHE");
    e->bellman4_policy_verbose = env_int("SL_BELLMAN4_POLICY_VERBOSE");
    e->bellman4_ragged = env_int("SL_BELLMAN4_RAGGED");
    e->bellman4_quarter = env_int("SL_BELLMAN4_QUARTER");
    e->bellman4_split = env_int("SL_BELLMAN4_SPLIT");
    e->bellman4_round = env_int("SL_BELLMAN4_ROUND");
    e->bellman4_shared = env_int("SL_BELLMAN4_SHARED");
    e->probe_blocks_per_cu = env_int("SL_PROBE_BLOCKS_PER_CU");
    e->succ_cache = env_int("SL_SUCC_CACHE");
}

extern "C" const char* sl_last_error(const sl_ctx* ctx) {
    return ctx ? ctx->error.c_str() : g_sl_last_error.c_str();
}

extern "C" int sl_ctx_create(int device, void* hip_stream, sl_ctx** out) {
    if (!out) return sl_fail(nullptr, SL_ERR_INVALID, "sl_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return sl_fail(nullptr, SL_ERR_HIP, "sl_ctx_create: no HIP device (%s)",
                       hipGetErrorString(e));
    if (device < 0 || device >= count)
        return sl_fail(nullptr, SL_ERR_INVALID, "sl_ctx_create: device %d out of range", device);
    sl_ctx* ctx = new (std::nothrow) sl_ctx();
    if (!ctx) return sl_fail(nullptr, SL_ERR_NOMEM, "sl_ctx_create: out of host memory");
    ctx->device = device;
    ctx->stream = (hipStream_t)hip_stream;
    memset(&ctx->h_model, 0, sizeof(ctx->h_model));
    memset(&ctx->h_gp, 0, sizeof(ctx->h_gp));
    memset(&ctx->h_tri, 0, sizeof(ctx->h_tri));
    memset(&ctx->h_net, 0, sizeof(ctx->h_net));
    SL_HIP_CHECK(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_gp, sizeof(SlGpDev)));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_tri, 2 * sizeof(SlTri)));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_net, sizeof(SlNet)));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_partials, sizeof(sl_key) * 4 * SL_MAX_GRID));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_partial_counts, sizeof(int64_t) * 2 * SL_MAX_GRID));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_ticket, 16));
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_actions, sizeof(double) * 1024));
    SL_HIP_CHECK(ctx, hipMemset(ctx->d_ticket, 0, 16));
    SL_HIP_CHECK(ctx, hipMemset(ctx->d_tri, 0, 2 * sizeof(SlTri)));
    SL_HIP_CHECK(ctx, hipMemset(ctx->d_net, 0, sizeof(SlNet)));
    SL_HIP_CHECK(ctx, hipMemset(ctx->d_gp, 0, sizeof(SlGpDev)));
    // environment switches are read HERE, once (not per launch)
    sl_env_read(&ctx->env);
    ctx->succ.enabled = ctx->env.succ_cache != 0;
    ctx->succ.max_bytes = -1;                      // default budget: a quarter of the device's memory
    *out = ctx;
    return SL_OK;
}

extern "C" int sl_ctx_destroy(sl_ctx* ctx) {
    if (!ctx) return SL_OK;
    (void)sl_comm_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int h = 0; h < SL_MAX_GP_HEADS; ++h) {
        if (ctx->gp_heads[h].d_xs) (void)hipFree(ctx->gp_heads[h].d_xs);
        if (ctx->gp_heads[h].d_mpack) (void)hipFree(ctx->gp_heads[h].d_mpack);
        if (ctx->gp_heads[h].d_alpha) (void)hipFree(ctx->gp_heads[h].d_alpha);
        if (ctx->gp_heads[h].d_kernel) (void)hipFree(ctx->gp_heads[h].d_kernel);
    }
    for (int s = 0; s < 2; ++s) if (ctx->d_tri_points[s]) (void)hipFree(ctx->d_tri_points[s]);
    if (ctx->d_net_kernels) (void)hipFree(ctx->d_net_kernels);
    (void)hipFree(ctx->d_gp);
    (void)hipFree(ctx->d_tri);
    (void)hipFree(ctx->d_net);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_gp4_seeds) (void)hipFree(ctx->d_gp4_seeds);
    if (ctx->d_policy_cache) (void)hipFree(ctx->d_policy_cache);
    if (ctx->d_pnet_params) (void)hipFree(ctx->d_pnet_params);
    if (ctx->d_policy_actions) (void)hipFree(ctx->d_policy_actions);
    if (ctx->succ.d) (void)hipFree(ctx->succ.d);
    if (ctx->succ.d_select) (void)hipFree(ctx->succ.d_select);
    if (ctx->d_records) (void)hipFree(ctx->d_records);
    (void)sl_timing_configure(ctx, 0);
    (void)hipFree(ctx->d_partials);
    (void)hipFree(ctx->d_partial_counts);
    (void)hipFree(ctx->d_ticket);
    (void)hipFree(ctx->d_actions);
    delete ctx;
    return SL_OK;
}

// ---- kernel timing for benchmarks ---------------------------------------------------------------
extern "C" int sl_timing_configure(sl_ctx* ctx, int slots) {
    if (!ctx || slots < 0) return sl_fail(ctx, SL_ERR_INVALID, "sl_timing_configure: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int ch = 0; ch < SL_TIMING_CHANNELS; ++ch) {
        if (ctx->timing.events[ch]) {
            for (int e = 0; e < 2 * ctx->timing.slots; ++e) (void)hipEventDestroy(ctx->timing.events[ch][e]);
            delete[] ctx->timing.events[ch];
            ctx->timing.events[ch] = nullptr;
        }
        ctx->timing.used[ch] = 0;
    }
    ctx->timing.slots = 0;
    if (!slots) return SL_OK;
    for (int ch = 0; ch < SL_TIMING_CHANNELS; ++ch) {
        ctx->timing.events[ch] = new (std::nothrow) hipEvent_t[2 * (size_t)slots]();
        if (!ctx->timing.events[ch]) return sl_fail(ctx, SL_ERR_NOMEM, "sl_timing_configure: out of host memory");
    }
    ctx->timing.slots = slots;
    for (int ch = 0; ch < SL_TIMING_CHANNELS; ++ch)
        for (int e = 0; e < 2 * slots; ++e) SL_HIP_CHECK(ctx, hipEventCreate(&ctx->timing.events[ch][e]));
    return SL_OK;
}

extern "C" int sl_timing_collect(sl_ctx* ctx, int channel, double* h_ms, int capacity, int* count) {
    if (!ctx || channel < 0 || channel >= SL_TIMING_CHANNELS || !count || capacity < 0 || (capacity && !h_ms))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_timing_collect: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int used = ctx->timing.used[channel];
    int n = 0;
    for (int s = 0; s < used && n < capacity; ++s, ++n) {
        float ms = 0.f;
        SL_HIP_CHECK(ctx, hipEventSynchronize(ctx->timing.events[channel][2 * s + 1]));
        SL_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->timing.events[channel][2 * s],
                                              ctx->timing.events[channel][2 * s + 1]));
        h_ms[n] = (double)ms;
    }
    *count = n;
    ctx->timing.used[channel] = 0;
    return SL_OK;
}

extern "C" int sl_ctx_synchronize(sl_ctx* ctx) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "NULL context");
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SL_OK;
}

static int sl_check_grid(sl_ctx* ctx, const sl_grid_desc& g, SlGridFast* gf) {
    if (g.d < 1 || g.d > SL_MAX_STATE_DIM)
        return sl_fail(ctx, SL_ERR_INVALID, "grid dimension %d outside [1,%d]", g.d,
                       SL_MAX_STATE_DIM);
    memset(gf, 0, sizeof(*gf));
    gf->d = g.d;
    gf->all_pow2 = 1;
    gf->nindex = 1;
    for (int k = 0; k < g.d; ++k) {
        int64_t n = g.num_points[k];
        if (n < 2) return sl_fail(ctx, SL_ERR_INVALID, "num_points[%d] = %lld < 2", k, (long long)n);
        if (gf->nindex > INT64_MAX / n) return sl_fail(ctx, SL_ERR_INVALID, "grid too large");
        gf->nindex *= n;
        gf->num32[k] = (uint32_t)n;
        if ((n & (n - 1)) == 0) {
            int s = 0;
            while ((1ll << s) < n) ++s;
            gf->shift[k] = s;
        } else {
            gf->all_pow2 = 0;
        }
    }
    return SL_OK;
}

extern "C" int sl_model_set(sl_ctx* ctx, const sl_model_desc* h_model) {
    if (!ctx || !h_model) return sl_fail(ctx, SL_ERR_INVALID, "sl_model_set: NULL argument");
    SlDevModel M;
    memset(&M, 0, sizeof(M));
    M.m = *h_model;
    int rc = sl_check_grid(ctx, M.m.grid, &M.gf);
    if (rc) return rc;
    const sl_policy_desc& p = M.m.policy;
    if (p.m < 1 || p.m > SL_MAX_ACTION_DIM)
        return sl_fail(ctx, SL_ERR_INVALID, "action dimension %d outside [1,%d]", p.m,
                       SL_MAX_ACTION_DIM);
    if (p.kind < SL_POLICY_LINEAR || p.kind > SL_POLICY_NETWORK)
        return sl_fail(ctx, SL_ERR_INVALID, "unknown policy kind %d", p.kind);
    if (p.kind == SL_POLICY_TABLE && !p.d_table)
        return sl_fail(ctx, SL_ERR_INVALID, "table policy without a table");
    M.in_dim = M.m.grid.d + p.m;
    if (M.in_dim > SL_MAX_INPUT_DIM)
        return sl_fail(ctx, SL_ERR_INVALID, "state+action dimension %d > %d", M.in_dim,
                       SL_MAX_INPUT_DIM);
    const int dk = M.m.dynamics.kind;
    if (dk < SL_DYN_LINEAR || dk > SL_DYN_GP)
        return sl_fail(ctx, SL_ERR_INVALID, "unknown dynamics kind %d", dk);
    if (dk == SL_DYN_PENDULUM && (M.m.grid.d != 2 || p.m != 1))
        return sl_fail(ctx, SL_ERR_INVALID, "pendulum dynamics need d=2, m=1");
    if (dk == SL_DYN_CARTPOLE && (M.m.grid.d != 4 || p.m != 1))
        return sl_fail(ctx, SL_ERR_INVALID, "cart-pole dynamics need d=4, m=1");
    M.uncertain = (dk == SL_DYN_GP);
    const int vk = M.m.value.kind;
    if (vk < SL_V_QUADRATIC || vk > SL_V_NETWORK)
        return sl_fail(ctx, SL_ERR_INVALID, "unknown value-function kind %d", vk);
    const int lk = M.m.lipschitz.lv_kind;
    if (lk < SL_LIP_CONST || lk > SL_LIP_NORM_GRAD)
        return sl_fail(ctx, SL_ERR_INVALID, "unknown L_v kind %d", lk);
    if (lk == SL_LIP_CONST || lk == SL_LIP_NORM_LINEAR || lk == SL_LIP_NORM_GRAD) M.m.lipschitz.lv_cols = 1;
    else M.m.lipschitz.lv_cols = M.m.grid.d;
    if ((lk == SL_LIP_ABS_GRAD || lk == SL_LIP_NORM_GRAD) && vk == SL_V_QUADRATIC)
        return sl_fail(ctx, SL_ERR_INVALID, "ABS_GRAD L_v needs a table or network V "
                                             "(use ABS_LINEAR with P + P^T)");
    // what k_bellman4_policy derives from the policy alone is kept between sweeps under this
    // token: a new policy description or another grid is a new policy (a model uploaded again
    // unchanged - every value-iteration sweep does that - is not)
    if (!ctx->model_set || memcmp(&ctx->h_model.m.policy, &M.m.policy, sizeof(M.m.policy)) != 0 ||
        memcmp(&ctx->h_model.m.grid, &M.m.grid, sizeof(M.m.grid)) != 0)
        ++ctx->policy_token;
    // the successor cache (sl_succ.hip) holds where f(x_i, u_a) lies in the value grid: another
    // grid, another dynamics description (kind, prior-mean / system matrix, normalisation) voids it
    if (!ctx->model_set || memcmp(&ctx->h_model.m.dynamics, &M.m.dynamics, sizeof(M.m.dynamics)) != 0 ||
        memcmp(&ctx->h_model.m.grid, &M.m.grid, sizeof(M.m.grid)) != 0 ||
        ctx->h_model.m.policy.m != M.m.policy.m)
        ++ctx->dynamics_token;
    // The model travels BY VALUE in every kernel's argument segment (sl_common.h): setting it is a
    // host-side copy - no device traffic, no synchronisation (a loop that uploads the same
    // description before every sweep, as the Python layer does, costs nothing here).
    ctx->h_model = M;
    ctx->model_set = true;
    return SL_OK;
}

// =============================================================================================
// auxiliary grids (Triangulation) and the network
// =============================================================================================
extern "C" int sl_tri_set(sl_ctx* ctx, int slot, const sl_grid_desc* h_grid, int nsimplex,
                          const int32_t* h_simplices, const double* h_hyperplanes,
                          const double* h_discrete_points, int project, int ncols,
                          const double* d_table) {
    if (!ctx || !h_grid || !h_simplices || !h_hyperplanes || !h_discrete_points)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_tri_set: NULL argument");
    if (slot < 0 || slot > 1) return sl_fail(ctx, SL_ERR_INVALID, "sl_tri_set: slot %d", slot);
    if (nsimplex < 1 || nsimplex > SL_MAX_SIMPLICES)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_tri_set: %d simplices (max %d)", nsimplex,
                       SL_MAX_SIMPLICES);
    SlGridFast gf;
    int rc = sl_check_grid(ctx, *h_grid, &gf);
    if (rc) return rc;
    if (ncols < 1) return sl_fail(ctx, SL_ERR_INVALID, "sl_tri_set: ncols < 1");
    SlTri& t = ctx->h_tri[slot];
    memset(&t, 0, sizeof(t));
    t.grid = *h_grid;
    t.nsimplex = nsimplex;
    t.project = project ? 1 : 0;
    t.ncols = ncols;
    t.set = 1;
    const int d = h_grid->d;
    for (int s = 0; s < nsimplex; ++s) {
        for (int v = 0; v <= d; ++v) t.simplices[s][v] = h_simplices[s * (d + 1) + v];
        for (int k = 0; k < d; ++k)
            for (int j = 0; j < d; ++j) t.hyper[s][k][j] = h_hyperplanes[(s * d + k) * d + j];
    }
    int64_t stride = 1, total = 0;
    for (int k = d - 1; k >= 0; --k) { t.stride[k] = stride; stride *= h_grid->num_points[k]; }
    for (int k = 0; k < d; ++k) { t.points_off[k] = (int32_t)total; total += h_grid->num_points[k]; }
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->d_tri_points[slot]) { (void)hipFree(ctx->d_tri_points[slot]); ctx->d_tri_points[slot] = nullptr; }
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_tri_points[slot], sizeof(double) * total));
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_tri_points[slot], h_discrete_points, sizeof(double) * total,
                                hipMemcpyHostToDevice));
    t.points = ctx->d_tri_points[slot];
    t.table = d_table;
    if (slot == 1) ++ctx->policy_token;
    if (slot == 0) ++ctx->dynamics_token;          // another value grid / simplices / projection
    sl_tri_finish(t, h_discrete_points);
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_tri + slot, &t, sizeof(SlTri), hipMemcpyHostToDevice));
    return SL_OK;
}

__global__ void k_set_table_pointer(SlTri* tri, const double* table) { tri->table = table; }

extern "C" int sl_tri_set_table(sl_ctx* ctx, int slot, const double* d_table) {
    if (!ctx || slot < 0 || slot > 1 || !ctx->h_tri[slot].set)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_tri_set_table: slot not set");
    ctx->h_tri[slot].table = d_table;
    if (slot == 1) ++ctx->policy_token;            // new vertex values of the policy
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // only the table pointer of the device descriptor changes: written by a one-thread kernel ON THE
    // STREAM, behind the sweeps that still read the old table and in front of those that follow -
    // no host synchronisation in a value-iteration loop that hands in a new table every sweep
    hipLaunchKernelGGL(k_set_table_pointer, dim3(1), dim3(1), 0, ctx->stream, ctx->d_tri + slot, d_table);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_network_set(sl_ctx* ctx, int nlayers, const int32_t* h_dims,
                              const int32_t* h_activations, const double* h_kernels) {
    if (!ctx || !h_dims || !h_activations || !h_kernels)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_network_set: NULL argument");
    if (nlayers < 1 || nlayers > SL_MAX_NN_LAYERS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_network_set: %d layers (max %d)", nlayers,
                       SL_MAX_NN_LAYERS);
    SlNet& n = ctx->h_net;
    memset(&n, 0, sizeof(n));
    n.nlayers = nlayers;
    n.set = 1;
    int64_t total = 0;
    for (int l = 0; l <= nlayers; ++l) {
        if (h_dims[l] < 1 || h_dims[l] > SL_NN_MAXW)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_network_set: width %d outside [1,%d]",
                           h_dims[l], SL_NN_MAXW);
        n.dims[l] = h_dims[l];
    }
    for (int l = 0; l < nlayers; ++l) {
        n.act[l] = h_activations[l];
        n.koff[l] = (int32_t)total;
        total += (int64_t)h_dims[l] * h_dims[l + 1];
    }
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->d_net_kernels) { (void)hipFree(ctx->d_net_kernels); ctx->d_net_kernels = nullptr; }
    std::vector<double> both(2 * (size_t)total);
    memcpy(both.data(), h_kernels, sizeof(double) * total);
    for (int l = 0; l < nlayers; ++l) {
        const int in = h_dims[l], out = h_dims[l + 1];
        const double* src = h_kernels + n.koff[l];
        double* dst = both.data() + total + n.koff[l];
        for (int o = 0; o < out; ++o)
            for (int i = 0; i < in; ++i) dst[(size_t)i * out + o] = src[(size_t)o * in + i];
    }
    // padded copies for the matrix-core kernels: rows = output features (multiple of 16),
    // row stride = padded input features + 2 (LDS bank spread)
    int64_t wtotal = 0;
    for (int l = 0; l < nlayers; ++l) {
        n.nfb[l] = (h_dims[l + 1] + 15) / 16;
        n.nib[l] = l == 0 ? (h_dims[0] + 15) / 16 : n.nfb[l - 1];
        n.nslab[l] = l == 0 ? (h_dims[0] + 3) / 4 : 4 * n.nfb[l - 1];
        n.wstride[l] = 16 * n.nib[l] + 2;
        n.woff[l] = (int32_t)wtotal;
        wtotal += (int64_t)16 * n.nfb[l] * n.wstride[l];
    }
    n.wtotal = (int32_t)wtotal;
    both.resize(2 * (size_t)total + (size_t)wtotal, 0.0);
    for (int l = 0; l < nlayers; ++l) {
        const int in = h_dims[l], out = h_dims[l + 1];
        const double* src = h_kernels + n.koff[l];
        double* dst = both.data() + 2 * total + n.woff[l];
        for (int o = 0; o < out; ++o)
            for (int i = 0; i < in; ++i) dst[(size_t)o * n.wstride[l] + i] = src[(size_t)o * in + i];
    }
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_net_kernels, sizeof(double) * both.size()));
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_net_kernels, both.data(), sizeof(double) * both.size(),
                                hipMemcpyHostToDevice));
    n.kernels = ctx->d_net_kernels;
    n.kernels_t = ctx->d_net_kernels + total;
    n.wpad = ctx->d_net_kernels + 2 * total;
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_net, &n, sizeof(SlNet), hipMemcpyHostToDevice));
    return SL_OK;
}

static int sl_check_ready(sl_ctx* ctx, const char* who) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "%s: NULL context", who);
    if (!ctx->model_set) return sl_fail(ctx, SL_ERR_INVALID, "%s: call sl_model_set first", who);
    const SlDevModel& M = ctx->h_model;
    if (M.m.value.kind == SL_V_TRI && !ctx->h_tri[0].set)
        return sl_fail(ctx, SL_ERR_INVALID, "%s: value table (sl_tri_set slot 0) not set", who);
    if (M.m.value.kind == SL_V_NETWORK && !ctx->h_net.set)
        return sl_fail(ctx, SL_ERR_INVALID, "%s: network not set", who);
    if (M.m.policy.kind == SL_POLICY_TRI && !ctx->h_tri[1].set)
        return sl_fail(ctx, SL_ERR_INVALID, "%s: policy table (sl_tri_set slot 1) not set", who);
    return SL_OK;
}

// =============================================================================================
// kernel-variant dispatch: fixed (state, action) dimensions fold every per-dimension predicate
// =============================================================================================
// variant ids: 0 = generic (dimensions read from the model), 1..4 = (d, 1) with d = 1..4
static inline int sl_dim_variant(const SlDevModel& M) {
    if (M.m.policy.m == 1 && M.m.grid.d >= 1 && M.m.grid.d <= 4) return M.m.grid.d;
    return 0;
}
#define SL_DISPATCH_DIMS(variant, general, CALL)                                     \
    do {                                                                             \
        if (general) {                                                               \
            if ((variant) == 2) { CALL(true, 2, 1); } else { CALL(true, 0, 0); }     \
        } else {                                                                     \
            switch (variant) {                                                       \
                case 1: CALL(false, 1, 1); break;                                    \
                case 2: CALL(false, 2, 1); break;                                    \
                case 3: CALL(false, 3, 1); break;                                    \
                case 4: CALL(false, 4, 1); break;                                    \
                default: CALL(false, 0, 0); break;                                   \
            }                                                                        \
        }                                                                            \
    } while (0)

// Wavefronts per SIMD the compiler has to leave room for (second argument of __launch_bounds__) in the
// GENERAL flavours - table V, interpolated policy - of the per-cell kernels below: left alone they
// take 256 + 126 ... 240 registers, one wavefront per SIMD (profiles/r06_gen_waves_ab.txt).
#ifndef SL_GEN_WAVES
#define SL_GEN_WAVES 2
#endif
// ... and the kernels that ONLY look a table up (k_values, k_policy_table of the general flavour with
// compile-time dimensions; the runtime-dimension fallbacks need the registers)
#ifndef SL_TABLE_WAVES
#define SL_TABLE_WAVES 4
#endif
// =============================================================================================
// values: V(x_i)                                             (lyapunov.py:305-322)
// =============================================================================================
template <bool GENERAL, int DT, int MT>
__global__ __launch_bounds__(SL_BLOCK, GENERAL ? (DT > 0 ? SL_TABLE_WAVES : SL_GEN_WAVES) : 1) void k_values(const SlDevModel M_arg, SlAux aux_arg, int64_t lo,
                                                     int64_t hi, double* __restrict__ values) {
    __shared__ SlTriLds<GENERAL> tri_lds;
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    SlDevModel M = M_arg;
    if (!GENERAL && DT > 0) {            // quadratic form and grid constants as vector operands
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            asm volatile("" : "+v"(M.m.grid.unit_maxes[k]));
            asm volatile("" : "+v"(M.m.grid.offset[k]));
            asm volatile("" : "+v"(M.m.grid.upper[k]));
#pragma unroll
            for (int q = 0; q < DT; ++q) asm volatile("" : "+v"(M.m.value.matrix[k][q]));
        }
    }
    const SlDims n = sl_dims<DT, MT>(M);
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P];
        sl_index_to_grid_point(M.m.grid, M.gf, n.d, idx, x);
        values[idx - lo] = sl_value_any<SlSweepFlavour<GENERAL>::value>(M, n.d, aux, x);
    }
}

extern "C" int sl_values(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_values) {
    int rc = sl_check_ready(ctx, "sl_values");
    if (rc) return rc;
    if (lo < 0 || hi < lo || hi > ctx->h_model.gf.nindex || !d_values)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_values: bad range or NULL output");
    if (hi == lo) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->h_model.m.value.kind == SL_V_NETWORK) return sl_nn_values_launch(ctx, lo, hi, d_values);
    SlAux aux{ctx->d_tri, ctx->d_net};
    const int blocks = sl_grid_blocks(hi - lo);
#define SL_CALL(G, D_, M_)                                                                    \
    hipLaunchKernelGGL((k_values<G, D_, M_>), dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream,   \
                       ctx->h_model, aux, lo, hi, d_values)
    SL_DISPATCH_DIMS(sl_dim_variant(ctx->h_model), sl_model_is_general(ctx->h_model), SL_CALL);
#undef SL_CALL
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// =============================================================================================
// table flavours on large training sets: action table, then k_gp_sweep4, then the check
// =============================================================================================
// A GP model whose V, L_v or policy is a piecewise-linear table cannot run inside k_gp_sweep4
// (its workgroups have no LDS left for the table descriptors); on more than 256 training points
// the sweep is therefore cut in three: the interpolated policy becomes a per-cell action table
// (k_policy_table), k_gp_sweep4 writes the posterior records of the closed loop with that table as
// its policy, and k_check_records runs the decrease check with the real V and L_v on the records
// (56 bytes per cell written and read back at d = 2, against ~0.5 MFLOP per cell at n = 512).
template <bool GENERAL, int DT, int MT>
__global__ __launch_bounds__(SL_BLOCK, GENERAL ? (DT > 0 ? SL_TABLE_WAVES : SL_GEN_WAVES) : 1) void k_policy_table(const SlDevModel M_arg, SlAux aux_arg,
                                                           int64_t lo, int64_t hi,
                                                           const double* __restrict__ points,
                                                           double* __restrict__ actions) {
    __shared__ SlTriLds<GENERAL> tri_lds;
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    const SlDevModel& M = M_arg;
    const SlDims n = sl_dims<DT, MT>(M);
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P], u[SL_M];
        sl_cell_state(M, n.d, idx, points, x);
        sl_policy_any<GENERAL>(M, n, aux.tri, idx, x, u);
#pragma unroll
        for (int a = 0; a < SL_M; ++a) if (a < n.m) actions[(idx - lo) * n.m + a] = u[a];
    }
}

// records: [decrease, threshold, mean[d], err[d]] per cell of [lo, hi) from the posterior-only pass
template <bool GENERAL, int DT, int MT>
__global__ __launch_bounds__(SL_BLOCK, GENERAL ? SL_GEN_WAVES : 1) void k_check_records(
    const SlDevModel M_arg, SlAux aux_arg, int64_t lo, int64_t hi, const uint64_t* __restrict__ init_bits,
    const double* __restrict__ values, const double* __restrict__ records,
    uint64_t* __restrict__ neg_bits, sl_key* __restrict__ partials, double* __restrict__ dbg,
    const double* __restrict__ points) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ SlTriLds<GENERAL> tri_lds;
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    const SlDevModel& M = M_arg;
    const SlDims n = sl_dims<DT, MT>(M);
    const int d = n.d;
    const int lane = threadIdx.x & 63;
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    for (int64_t base = lo + (int64_t)blockIdx.x * SL_BLOCK; base < hi;
         base += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t idx = base + threadIdx.x;
        const bool valid = idx < hi;
        const int64_t wbase = base + (threadIdx.x & ~63);        // first cell of this wavefront
        const uint64_t init = (init_bits && wbase < hi) ? init_bits[(wbase - lo) >> 6] : 0ull;
        bool negative = false;
        double v_x = 0.0;
        if (valid) {
            double x[SL_P], u[SL_M], mean[SL_D], err[SL_D];
            const double* rec = records + (idx - lo) * (2 + 2 * d);
#pragma unroll
            for (int k = 0; k < SL_D; ++k) if (k < d) { mean[k] = rec[2 + k]; err[k] = rec[2 + d + k]; }
            sl_cell_state(M, d, idx, points, x);
            sl_policy_any<GENERAL>(M, n, aux.tri, idx, x, u);
            sl_append_action(n, u, x);
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
        const uint64_t word = __ballot(negative);
        if (wbase < hi) {
            if (lane == 0) neg_bits[(wbase - lo) >> 6] = word;
            const bool ok = negative || ((init >> lane) & 1ull);
            if (valid && !ok) sl_key_min(best_v, best_i, sl_vbits(v_x), idx);
        }
    }
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (threadIdx.x == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// =============================================================================================
// deterministic-dynamics decrease check                    (lyapunov.py:436-441, 524-535)
// =============================================================================================
// The model constants of the closed-form path outnumber the scalar registers (the compiler spills
// them into VGPR lanes and reads every use back with v_readlane); holding each one in its own
// VGPR - same value in all lanes, opaque to the optimiser - makes them plain vector operands.
template <int DT, int MT, int DYN>
__device__ __forceinline__ void sl_constants_to_vgprs(SlDevModel& L) {
#define SL_TO_VGPR(x) asm volatile("" : "+v"(x))
    if (DYN != SL_DYN_LINEAR) {
#pragma unroll
        for (int q = 0; q < 16; ++q) SL_TO_VGPR(L.m.dynamics.coef[q]);
#pragma unroll
        for (int k = 0; k < DT; ++k) { SL_TO_VGPR(L.m.dynamics.tx[k]); SL_TO_VGPR(L.m.dynamics.tx_inv[k]); }
#pragma unroll
        for (int a = 0; a < MT; ++a) SL_TO_VGPR(L.m.dynamics.tu[a]);
    }
#pragma unroll
    for (int k = 0; k < DT; ++k) {
        SL_TO_VGPR(L.m.grid.unit_maxes[k]);
        SL_TO_VGPR(L.m.grid.offset[k]);
        if (DYN == SL_DYN_LINEAR) {
#pragma unroll
            for (int q = 0; q < DT + MT; ++q) SL_TO_VGPR(L.m.dynamics.matrix[k][q]);
        }
#pragma unroll
        for (int q = 0; q < DT; ++q) {
            SL_TO_VGPR(L.m.value.matrix[k][q]);
            SL_TO_VGPR(L.m.lipschitz.lv_matrix[k][q]);
        }
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int k = 0; k < DT; ++k) SL_TO_VGPR(L.m.policy.matrix[a][k]);
        SL_TO_VGPR(L.m.policy.lower[a]);
        SL_TO_VGPR(L.m.policy.upper[a]);
    }
    SL_TO_VGPR(L.m.lipschitz.lv_const);
    SL_TO_VGPR(L.m.lipschitz.lf_const);
    SL_TO_VGPR(L.m.lipschitz.tau);
#undef SL_TO_VGPR
}

// POW2: every grid axis has a power-of-two length and the grid fewer than 2^32 cells (128^4):
// the cell state comes from 32-bit shifts and masks held in vector registers - the generic decode
// (three code paths, 64-bit shifts, operands read back from spilled scalar registers with
// v_readlane) cost as much as the FP64 arithmetic of the check.  No explicit points, no records.
template <bool GENERAL, int DT, int MT, int DYN, bool POW2 = false>
__global__ __launch_bounds__(SL_BLOCK, GENERAL ? SL_GEN_WAVES : 1) void k_det_sweep(
    const SlDevModel M_arg, SlAux aux_arg, int64_t lo, int64_t hi, const uint64_t* __restrict__ init_bits,
    const double* __restrict__ values, uint64_t* __restrict__ neg_bits,
    sl_key* __restrict__ partials, double* __restrict__ dbg, const double* __restrict__ points) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ SlTriLds<GENERAL> tri_lds;
    const SlAux aux = sl_stage_aux<GENERAL>(tri_lds, aux_arg);
    SlDevModel M = M_arg;
    if (!GENERAL && DT > 0 && DYN != 0) sl_constants_to_vgprs<DT, MT, DYN>(M);
    uint32_t axis_mask[DT > 0 ? DT : 1], axis_shift[DT > 0 ? DT : 1];
    if (POW2) {
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            axis_shift[k] = (uint32_t)M.gf.shift[k];
            axis_mask[k] = (1u << axis_shift[k]) - 1u;
            asm volatile("" : "+v"(axis_mask[k]));
            asm volatile("" : "+v"(axis_shift[k]));
        }
        // The check is one long chain of dependent (unfused) FP64 operations per cell and the
        // ~70 vector-resident constants leave room for two wavefronts per SIMD only: every
        // thread of the linear-dynamics variant handles TWO cells (cell and cell + 64) whose chains
        // interleave (the Euler variants have no registers left for that: one cell).  No branches
        // on `valid`: out-of-range lanes compute the last cell and are masked afterwards.
        constexpr int CPT = (DYN == SL_DYN_LINEAR) ? 2 : 1;
        const SlDims n2 = sl_dims<DT, MT>(M);
        const int lane2 = threadIdx.x & 63;
        uint64_t bv = ~0ull;
        int64_t bi = INT64_MAX;
        const int64_t wave_off = (int64_t)(threadIdx.x >> 6) * (64 * CPT);
        for (int64_t base = lo + (int64_t)blockIdx.x * (CPT * SL_BLOCK); base < hi;
             base += (int64_t)gridDim.x * (CPT * SL_BLOCK)) {
            const int64_t wbase = base + wave_off;               // first of this wavefront's 64 CPT cells
            bool neg2[CPT];
            double vx2[CPT];
            int64_t idx2[CPT];
            uint64_t init2[CPT];
            // everything this iteration reads from memory is requested first: the ~1000 cycles
            // of arithmetic below hide the latency (measured before: 53 % of the wave time
            // in s_waitcnt on the initial-set word, which was loaded after the ballot)
            // (the Euler variants keep their loads behind the long arithmetic: requesting them
            // first costs registers there - 10.7 -> 13.2 ms at 128^4)
            if (CPT > 1) {
#pragma unroll
                for (int t = 0; t < CPT; ++t) {
                    const int64_t w0 = wbase + 64 * t;
                    const int64_t raw = w0 + lane2;
                    const int64_t idx = raw < hi ? raw : hi - 1;
                    init2[t] = (init_bits && w0 < hi) ? init_bits[(w0 - lo) >> 6] : 0ull;
                    vx2[t] = values ? values[idx - lo] : 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < CPT; ++t) {
                const int64_t raw = wbase + 64 * t + lane2;
                const int64_t idx = raw < hi ? raw : hi - 1;
                idx2[t] = raw;
                double x[SL_P], u[SL_M], nxt[SL_D], err[SL_D];
                uint32_t r = (uint32_t)idx;
#pragma unroll
                for (int k = DT - 1; k >= 0; --k) {
                    const int ijk = (int)(r & axis_mask[k]);
                    r >>= axis_shift[k];
                    const double tt = (double)ijk * M.m.grid.unit_maxes[k];     // functions.py:731
                    x[k] = tt + M.m.grid.offset[k];
                }
                sl_policy_any<GENERAL>(M, n2, aux.tri, idx, x, u);
                sl_append_action(n2, u, x);
                sl_dynamics_det<DYN>(M, n2, x, nxt);
                SlCellCheck c = sl_cell_check<SlSweepFlavour<GENERAL>::value>(M, n2.d, aux, x, nxt, err);
                neg2[t] = c.negative && raw < hi;
                if (!values) vx2[t] = c.v_x;                     // ordering key: lyapunov.py:512
                else if (CPT == 1) vx2[t] = values[idx - lo];
                if (CPT == 1) init2[t] = (init_bits && wbase < hi) ? init_bits[(wbase - lo) >> 6] : 0ull;
            }
#pragma unroll
            for (int t = 0; t < CPT; ++t) {
                const uint64_t word = __ballot(neg2[t]);
                const int64_t w0 = wbase + 64 * t;
                if (w0 < hi) {
                    const int64_t widx = (w0 - lo) >> 6;
                    if (lane2 == 0) neg_bits[widx] = word;
                    const bool ok = neg2[t] || ((init2[t] >> lane2) & 1ull);
                    if (idx2[t] < hi && !ok) sl_key_min(bv, bi, sl_vbits(vx2[t]), idx2[t]);
                }
            }
        }
        sl_block_reduce_key<true>(bv, bi, sv, si);
        if (threadIdx.x == 0) { partials[blockIdx.x].vbits = bv; partials[blockIdx.x].index = bi; }
        return;
    }
    const SlDims n = sl_dims<DT, MT>(M);
    const int d = n.d;
    const int lane = threadIdx.x & 63;
    uint64_t best_v = ~0ull;
    int64_t best_i = INT64_MAX;
    for (int64_t base = lo + (int64_t)blockIdx.x * SL_BLOCK; base < hi;
         base += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t idx = base + threadIdx.x;
        const bool valid = idx < hi;
        bool negative = false;
        double v_x = 0.0;
        // the initial-set word of this wavefront is requested before the arithmetic, not after
        // the ballot (its latency was fully exposed there)
        const int64_t wbase = base + (threadIdx.x & ~63);        // first cell of this wavefront
        const uint64_t init = (init_bits && wbase < hi) ? init_bits[(wbase - lo) >> 6] : 0ull;
        if (valid) {
            double x[SL_P], u[SL_M], nxt[SL_D], err[SL_D];
            sl_cell_state(M, d, idx, points, x);
            sl_policy_any<GENERAL>(M, n, aux.tri, idx, x, u);
            sl_append_action(n, u, x);
            sl_dynamics_det<DYN>(M, n, x, nxt);
            SlCellCheck c = sl_cell_check<SlSweepFlavour<GENERAL>::value>(M, d, aux, x, nxt, err);
            negative = c.negative;
            v_x = values ? values[idx - lo] : c.v_x;       // ordering key: lyapunov.py:512
            if (dbg) {
                double* o = dbg + (idx - lo) * (2 + 2 * d);
                o[0] = c.decrease; o[1] = c.threshold;
#pragma unroll
                for (int k = 0; k < SL_D; ++k) if (k < d) { o[2 + k] = nxt[k]; o[2 + d + k] = 0.0; }
            }
        }
        const uint64_t word = __ballot(negative);
        if (wbase < hi) {
            const int64_t widx = (wbase - lo) >> 6;
            if (lane == 0) neg_bits[widx] = word;
            const bool ok = negative || ((init >> lane) & 1ull);
            if (valid && !ok) sl_key_min(best_v, best_i, sl_vbits(v_x), idx);
        }
    }
    sl_block_reduce_key<true>(best_v, best_i, sv, si);
    if (threadIdx.x == 0) { partials[blockIdx.x].vbits = best_v; partials[blockIdx.x].index = best_i; }
}

// partials[0..n) -> result->fail
__global__ __launch_bounds__(SL_BLOCK) void k_reduce_fail(const sl_key* __restrict__ partials,
                                                          int n, sl_sweep_result* result) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    uint64_t v = ~0ull;
    int64_t i = INT64_MAX;
    for (int k = threadIdx.x; k < n; k += SL_BLOCK) sl_key_min(v, i, partials[k].vbits, partials[k].index);
    sl_block_reduce_key<true>(v, i, sv, si);
    if (threadIdx.x == 0) { result->fail.vbits = v; result->fail.index = i; }
}


// shared by sl_lyap_sweep (grid cells) and sl_eval_points (explicit points)
// the model the posterior pass of a split sweep sees: same grid, policy and GP; a zero quadratic V
// and a scalar L_v (the fast path of the GP kernels) - its records carry mean and error only
static SlDevModel sl_posterior_only(const SlDevModel& full) {
    SlDevModel m = full;
    memset(&m.m.value, 0, sizeof(m.m.value));
    m.m.value.kind = SL_V_QUADRATIC;
    m.m.lipschitz.lv_kind = SL_LIP_CONST;
    m.m.lipschitz.lv_cols = 1;
    return m;
}

// table flavours (V, L_v = |grad V|, interpolated policy) of a GP model whose heads are served by
// k_gp_sweep4: action table + posterior records + check instead of k_gp_sweep's 16x16x4 structure
static bool sl_gp_three_pass(sl_ctx* ctx) {
    const SlDevModel& M = ctx->h_model;
    if (M.m.value.kind == SL_V_NETWORK || !sl_model_is_general(M)) return false;
    bool other_kernels = false;
    for (int h = 0; h < ctx->h_gp.nheads; ++h) other_kernels = other_kernels || ctx->gp_heads[h].d_kernel;
    SlDevModel po = sl_posterior_only(M);
    if (po.m.policy.kind == SL_POLICY_TRI) po.m.policy.kind = SL_POLICY_TABLE;
    if (ctx->gp_cfg == 2 && !other_kernels && sl_gp4_supports(po)) return true;
    // small training sets (k_gp_small, the notebooks' regime): the same split
    if (ctx->env.gp_small_split == 1 && (other_kernels || ctx->gp_cfg == 0) && sl_gp_small_supports(ctx, po))
        return true;
    return false;
}

int sl_sweep_any(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                 const double* d_values, uint64_t* d_neg_bits, sl_sweep_result* d_result,
                 double* d_dbg, const double* d_points) {
    int rc = sl_check_ready(ctx, "sl_lyap_sweep");
    if (rc) return rc;
    if (lo < 0 || hi < lo || (!d_points && hi > ctx->h_model.gf.nindex) ||
        ((lo & 63) && hi != lo))      // an empty shard may start anywhere (tail ranks of a small grid)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_lyap_sweep: bad range (lo must be a multiple of 64)");
    // with explicit points a TABLE policy is indexed by the point number (one action per point)
    if (!d_neg_bits || !d_result) return sl_fail(ctx, SL_ERR_INVALID, "sl_lyap_sweep: NULL output");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SlPolicyTableScope network_policy(ctx, lo, hi, d_points);   // (a network policy becomes a per-cell table)
    if (network_policy.rc) return network_policy.rc;
    int blocks = 1;
    ctx->last_kernel[0] = 0;
    if (hi == lo) {
        blocks = 0;
    } else if (ctx->h_model.m.value.kind == SL_V_NETWORK) {
        // network V: (1) GP posterior records from the MFMA kernel, (2) cooperative network check
        const double* records = nullptr;
        if (ctx->h_model.m.dynamics.kind == SL_DYN_GP) {
            const int d = ctx->h_model.m.grid.d;
            const size_t need = sizeof(double) * (size_t)(hi - lo) * (2 + 2 * d) +
                                sizeof(uint64_t) * (size_t)((hi - lo + 63) / 64 + 1);
            if (need > ctx->records_bytes) {
                if (ctx->d_records) (void)hipFree(ctx->d_records);
                ctx->d_records = nullptr;
                ctx->records_bytes = 0;
                SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_records, need));
                ctx->records_bytes = need;
            }
            double* rec = reinterpret_cast<double*>(ctx->d_records);
            uint64_t* tmp_bits = reinterpret_cast<uint64_t*>(rec + (size_t)(hi - lo) * (2 + 2 * d));
            SlDevModel posterior_only = sl_posterior_only(ctx->h_model);
            int gp_blocks = 0;
            rc = sl_gp_sweep_launch(ctx, posterior_only, lo, hi, nullptr, nullptr, tmp_bits,
                                    &gp_blocks, rec, d_points);
            if (rc) return rc;
            records = rec;
        }
        rc = sl_nn_check_launch(ctx, lo, hi, d_init_bits, d_values, records, d_neg_bits, &blocks,
                                d_dbg, d_points);
        if (rc) return rc;
    } else if (ctx->h_model.m.dynamics.kind == SL_DYN_GP && sl_gp_three_pass(ctx)) {
        // table V / L_v = |grad V| / interpolated policy on a large training set (see k_policy_table)
        const SlDevModel& full = ctx->h_model;
        const int d = full.m.grid.d, m = full.in_dim - d;
        const size_t rec_doubles = (size_t)(hi - lo) * (2 + 2 * d), act_doubles = (size_t)(hi - lo) * m;
        const size_t need = sizeof(double) * (rec_doubles + act_doubles) +
                            sizeof(uint64_t) * (size_t)((hi - lo + 63) / 64 + 1);
        if (need > ctx->records_bytes) {
            if (ctx->d_records) (void)hipFree(ctx->d_records);
            ctx->d_records = nullptr;
            ctx->records_bytes = 0;
            SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_records, need));
            ctx->records_bytes = need;
        }
        double* rec = reinterpret_cast<double*>(ctx->d_records);
        double* act = rec + rec_doubles;
        uint64_t* tmp_bits = reinterpret_cast<uint64_t*>(act + act_doubles);
        SlAux aux{ctx->d_tri, ctx->d_net};
        SlDevModel posterior_only = sl_posterior_only(full);
        const int variant = sl_dim_variant(full);
        const bool tri_policy = full.m.policy.kind == SL_POLICY_TRI;
        blocks = sl_grid_blocks(hi - lo);
        if (tri_policy) {
#define SL_CALL(G, D_, M_)                                                                     \
    hipLaunchKernelGGL((k_policy_table<true, D_, M_>), dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, \
                       full, aux, lo, hi, d_points, act)
            SL_DISPATCH_DIMS(variant, true, SL_CALL);
#undef SL_CALL
            SL_HIP_CHECK(ctx, hipGetLastError());
            // the per-cell table is indexed by the cell (or point) number
            posterior_only.m.policy.kind = SL_POLICY_TABLE;
            posterior_only.m.policy.d_table = act - lo * m;
        }
        int gp_blocks = 0;
        rc = sl_gp_sweep_launch(ctx, posterior_only, lo, hi, nullptr, nullptr, tmp_bits, &gp_blocks,
                                rec, d_points);
        if (rc) return rc;
#define SL_CALL(G, D_, M_)                                                                     \
    hipLaunchKernelGGL((k_check_records<true, D_, M_>), dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, \
                       full, aux, lo, hi, d_init_bits, d_values, rec, d_neg_bits, ctx->d_partials, \
                       d_dbg, d_points)
        SL_DISPATCH_DIMS(variant, true, SL_CALL);
#undef SL_CALL
        SL_HIP_CHECK(ctx, hipGetLastError());
        sl_note_kernel(ctx, true, tri_policy ? "k_policy_table + k_check_records" : "k_check_records");
    } else if (ctx->h_model.m.dynamics.kind == SL_DYN_GP) {
        rc = sl_gp_sweep_launch(ctx, ctx->h_model, lo, hi, d_init_bits, d_values, d_neg_bits,
                                &blocks, d_dbg, d_points);
        if (rc) return rc;
    } else if (!d_dbg && !d_points && ctx->env.det_rows != 0 && sl_det_rows_supports(ctx->h_model, lo, hi)) {
        // linear dynamics / linear policy / quadratic V: 8 cells of a grid row per thread
        rc = sl_det_rows_launch(ctx, lo, hi, d_init_bits, d_values, d_neg_bits, &blocks);
        if (rc) return rc;
    } else {
        blocks = sl_grid_blocks(hi - lo);
        SlAux aux{ctx->d_tri, ctx->d_net};
        const int dyn = ctx->h_model.m.dynamics.kind;
        const bool pow2 = ctx->h_model.gf.all_pow2 && ctx->h_model.gf.nindex <= 0xffffffffll &&
                          !d_dbg && !d_points;
#define SL_LAUNCH_DET(G, D_, M_, DYN_)                                                          \
    do {                                                                                        \
        if (pow2 && !(G) && (D_) > 0 && (DYN_) != 0)                                            \
            hipLaunchKernelGGL((k_det_sweep<G, D_, M_, DYN_, (!(G) && (D_) > 0 && (DYN_) != 0)>), \
                               dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, ctx->h_model, aux, \
                               lo, hi, d_init_bits, d_values, d_neg_bits, ctx->d_partials,      \
                               d_dbg, d_points);                                                \
        else                                                                                    \
            hipLaunchKernelGGL((k_det_sweep<G, D_, M_, DYN_>), dim3(blocks), dim3(SL_BLOCK), 0, \
                               ctx->stream, ctx->h_model, aux, lo, hi, d_init_bits, d_values,  \
                               d_neg_bits, ctx->d_partials, d_dbg, d_points);                   \
    } while (0)
#define SL_CALL(G, D_, M_)                                                                     \
    do {                                                                                       \
        if (!(G) && (D_) > 0 && dyn == SL_DYN_LINEAR) SL_LAUNCH_DET(G, D_, M_, SL_DYN_LINEAR); \
        else if (!(G) && (D_) == 2 && dyn == SL_DYN_PENDULUM) SL_LAUNCH_DET(G, D_, M_, SL_DYN_PENDULUM); \
        else if (!(G) && (D_) == 4 && dyn == SL_DYN_CARTPOLE) SL_LAUNCH_DET(G, D_, M_, SL_DYN_CARTPOLE); \
        else SL_LAUNCH_DET(G, D_, M_, 0);                                                      \
    } while (0)
        SL_DISPATCH_DIMS(sl_dim_variant(ctx->h_model), sl_model_is_general(ctx->h_model), SL_CALL);
        sl_note_kernel(ctx, false, "k_det_sweep<general=%d, d=%d, dynamics=%d, pow2=%d>",
                       (int)sl_model_is_general(ctx->h_model), sl_dim_variant(ctx->h_model), dyn, (int)pow2);
#undef SL_CALL
#undef SL_LAUNCH_DET
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_reduce_fail, dim3(1), dim3(SL_BLOCK), 0, ctx->stream, ctx->d_partials,
                       blocks, d_result);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_lyap_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                             const double* d_values, uint64_t* d_neg_bits,
                             sl_sweep_result* d_result, double* d_dbg) {
    SlTimed timed(ctx, 0);
    return sl_sweep_any(ctx, lo, hi, d_init_bits, d_values, d_neg_bits, d_result, d_dbg, nullptr);
}

// =============================================================================================
// finalisation: the prefix rule of lyapunov.py:513-606 in parallel form
// =============================================================================================
__global__ __launch_bounds__(SL_BLOCK) void k_finalize(
    int64_t lo, int64_t hi, const double* __restrict__ values,
    const uint64_t* __restrict__ init_bits, const uint64_t* __restrict__ prev_bits, sl_key star,
    sl_key keep, uint64_t* __restrict__ safe_bits, sl_key* __restrict__ partials,
    int64_t* __restrict__ counts) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ int64_t sc[2][SL_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ls_v = 0ull, mx_v = 0ull;
    int64_t ls_i = -1, mx_i = -1;
    int64_t n_below = 0, n_safe = 0;
    for (int64_t base = lo + (int64_t)blockIdx.x * SL_BLOCK; base < hi;
         base += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t idx = base + threadIdx.x;
        const bool valid = idx < hi;
        const int64_t wbase = base + (threadIdx.x & ~63);
        bool safe = false;
        if (wbase < hi) {
            const int64_t widx = (wbase - lo) >> 6;
            const uint64_t init = init_bits ? init_bits[widx] : 0ull;
            const uint64_t prev = prev_bits ? prev_bits[widx] : 0ull;
            if (valid) {
                const uint64_t vb = sl_vbits(values[idx - lo]);
                const bool below = sl_key_less(vb, idx, star.vbits, star.index);
                const bool kept = ((prev >> lane) & 1ull) && !sl_key_less(vb, idx, keep.vbits, keep.index);
                safe = below || kept || ((init >> lane) & 1ull);
                if (below) { ++n_below; sl_key_max(ls_v, ls_i, vb, idx); }
                sl_key_max(mx_v, mx_i, vb, idx);
            }
            const uint64_t word = __ballot(safe);
            if (lane == 0) { safe_bits[widx] = word; n_safe += __popcll(word); }
        }
    }
    // block reductions
    for (int off = 32; off >= 1; off >>= 1) {
        n_below += __shfl_xor((long long)n_below, off, 64);
        n_safe += __shfl_xor((long long)n_safe, off, 64);
    }
    if (lane == 0) { sc[0][wave] = n_below; sc[1][wave] = n_safe; }
    sl_block_reduce_key<false>(ls_v, ls_i, sv, si);
    __syncthreads();
    sl_block_reduce_key<false>(mx_v, mx_i, sv, si);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x].vbits = ls_v; partials[2 * blockIdx.x].index = ls_i;
        partials[2 * blockIdx.x + 1].vbits = mx_v; partials[2 * blockIdx.x + 1].index = mx_i;
        int64_t a = 0, b = 0;
        for (int w = 0; w < SL_BLOCK / 64; ++w) { a += sc[0][w]; b += sc[1][w]; }
        counts[2 * blockIdx.x] = a; counts[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_reduce_finalize(const sl_key* __restrict__ partials,
                                                              const int64_t* __restrict__ counts,
                                                              int n, sl_sweep_result* result) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ int64_t sc[2][SL_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ls_v = 0ull, mx_v = 0ull;
    int64_t ls_i = -1, mx_i = -1, a = 0, b = 0;
    for (int k = threadIdx.x; k < n; k += SL_BLOCK) {
        sl_key_max(ls_v, ls_i, partials[2 * k].vbits, partials[2 * k].index);
        sl_key_max(mx_v, mx_i, partials[2 * k + 1].vbits, partials[2 * k + 1].index);
        a += counts[2 * k]; b += counts[2 * k + 1];
    }
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor((long long)a, off, 64);
        b += __shfl_xor((long long)b, off, 64);
    }
    if (lane == 0) { sc[0][wave] = a; sc[1][wave] = b; }
    sl_block_reduce_key<false>(ls_v, ls_i, sv, si);
    __syncthreads();
    sl_block_reduce_key<false>(mx_v, mx_i, sv, si);
    if (threadIdx.x == 0) {
        result->last_safe.vbits = ls_v; result->last_safe.index = ls_i;
        result->max_key.vbits = mx_v; result->max_key.index = mx_i;
        int64_t ta = 0, tb = 0;
        for (int w = 0; w < SL_BLOCK / 64; ++w) { ta += sc[0][w]; tb += sc[1][w]; }
        result->count_below = ta; result->count_safe = tb;
    }
}

extern "C" int sl_lyap_finalize(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                                const uint64_t* d_init_bits, const uint64_t* d_prev_bits,
                                sl_key key_star, sl_key key_keep, uint64_t* d_safe_bits,
                                sl_sweep_result* d_result) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_lyap_finalize: NULL context");
    if (lo < 0 || hi < lo || ((lo & 63) && hi != lo) || !d_values || !d_safe_bits || !d_result)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_lyap_finalize: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int blocks = (hi == lo) ? 0 : sl_grid_blocks(hi - lo);
    if (blocks) {
        hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, lo, hi,
                           d_values, d_init_bits, d_prev_bits, key_star, key_keep, d_safe_bits,
                           ctx->d_partials, ctx->d_partial_counts);
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_reduce_finalize, dim3(1), dim3(SL_BLOCK), 0, ctx->stream, ctx->d_partials,
                       ctx->d_partial_counts, blocks, d_result);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// =============================================================================================
// radix-select histogram pass over (V, index) keys
// =============================================================================================
__global__ __launch_bounds__(SL_BLOCK) void k_select_pass(int64_t lo, int64_t hi,
                                                          const double* __restrict__ values,
                                                          int which, int byte, uint64_t prefix,
                                                          uint64_t vbits_equal,
                                                          uint64_t* __restrict__ hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const int shift = byte * 8;
    const uint64_t himask = (byte == 7) ? 0ull : (~0ull << (shift + 8));
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        const uint64_t vb = sl_vbits(values[idx - lo]);
        uint64_t key;
        bool take;
        if (which == 0) { key = vb; take = true; }
        else { key = (uint64_t)idx; take = (vb == vbits_equal); }
        take = take && ((key & himask) == (prefix & himask));
        if (take) atomicAdd(&lh[(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    const unsigned int c = lh[threadIdx.x];
    if (c) atomicAdd((unsigned long long*)&hist[threadIdx.x], (unsigned long long)c);
}

extern "C" int sl_select_pass(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                              int which, int byte, uint64_t prefix, uint64_t vbits_equal,
                              uint64_t* d_hist) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_select_pass: NULL context");
    if (lo < 0 || hi < lo || !d_values || !d_hist || byte < 0 || byte > 7 || which < 0 || which > 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_select_pass: bad argument");
    if (hi == lo) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // a block must not overflow its 32-bit local counters: <= 2^31 cells per block
    hipLaunchKernelGGL(k_select_pass, dim3(sl_grid_blocks(hi - lo)), dim3(SL_BLOCK), 0,
                       ctx->stream, lo, hi, d_values, which, byte, prefix, vbits_equal, d_hist);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// =============================================================================================
// bit mask <-> bool[N] byte mask
// =============================================================================================
__global__ __launch_bounds__(SL_BLOCK) void k_bits_to_bytes(int64_t n, const uint64_t* __restrict__ bits,
                                                            uint8_t* __restrict__ bytes) {
    // one thread per 8 cells: 8-byte store
    const int64_t ngroups = (n + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; g < ngroups;
         g += (int64_t)gridDim.x * SL_BLOCK) {
        const uint64_t word = bits[g >> 3];
        const unsigned int b = (unsigned int)((word >> ((g & 7) * 8)) & 0xff);
        // spread 8 bits into 8 bytes
        uint64_t x = b;
        x = (x | (x << 28)) & 0x0000000f0000000full;
        x = (x | (x << 14)) & 0x0003000300030003ull;
        x = (x | (x << 7)) & 0x0101010101010101ull;
        if (g * 8 + 8 <= n) {
            *reinterpret_cast<uint64_t*>(bytes + g * 8) = x;
        } else {
            for (int k = 0; g * 8 + k < n; ++k) bytes[g * 8 + k] = (uint8_t)((x >> (8 * k)) & 1);
        }
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_bytes_to_bits(int64_t n, const uint8_t* __restrict__ bytes,
                                                            uint64_t* __restrict__ bits) {
    for (int64_t base = (int64_t)blockIdx.x * SL_BLOCK; base < n; base += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t idx = base + threadIdx.x;
        const bool on = (idx < n) && bytes[idx] != 0;
        const uint64_t word = __ballot(on);
        const int64_t wbase = base + (threadIdx.x & ~63);
        if ((threadIdx.x & 63) == 0 && wbase < n) bits[wbase >> 6] = word;
    }
}

extern "C" int sl_bits_to_bytes(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, uint8_t* d_bytes) {
    if (!ctx || n < 0 || !d_bits || !d_bytes)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bits_to_bytes: bad argument");
    if (n == 0) return SL_OK;
    if (((uintptr_t)d_bytes) & 7)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bits_to_bytes: d_bytes must be 8-byte aligned");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_bits_to_bytes, dim3(sl_grid_blocks((n + 7) / 8)), dim3(SL_BLOCK), 0,
                       ctx->stream, n, d_bits, d_bytes);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_bytes_to_bits(sl_ctx* ctx, int64_t n, const uint8_t* d_bytes, uint64_t* d_bits) {
    if (!ctx || n < 0 || !d_bits || !d_bytes)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bytes_to_bits: bad argument");
    if (n == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_bytes_to_bits, dim3(sl_grid_blocks(n)), dim3(SL_BLOCK), 0, ctx->stream, n,
                       d_bytes, d_bits);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}
