// sl_region.hip - get_lyapunov_region (lyapunov.py:59-139) in parallel form.
//
// The reference floods the grid from a start node with a priority queue keyed on the function
// value: the smallest frontier node is popped next; the flood ends when the popped node lies on the
// grid boundary (it is dropped) or is LOWER than the one before (the water would run downhill into
// another basin; that node is kept), and frontier nodes that were never popped are dropped.
//
// As long as the popped values do not decrease, the set popped before any node of value >= c is the
// connected component (3^d - 1 neighbours) of the start node in {V < c}.  With the minimax distance
//     D(v) = min over paths start -> v of the largest value on the path (v included),
// the flood therefore ends at the level
//     c* = min( min { D(b) : b on the grid boundary },  min { D(y) : D(y) > V(y) } )
// (the first boundary node, or the first node that can only be reached by descending), the last
// regular pop is the node x* with V(x*) = D(x*) = c*, and the region is
//     { D(v) < c* }  +  x*  +  the lowest neighbour y of x* with V(y) < c* = D(y)     (descent), or
//     { D(v) < c* }                                                                  (x* on the boundary;
//                                                       a descent ONTO the boundary drops y as well).
// D is the fixpoint of D(v) = min(D(v), max(V(v), min_u D(u))) from D(start) = V(start): relaxed in
// place by full-grid passes (monotone: any order, races included, converges to the same fixpoint),
// pruned by the lowest boundary value reached so far (tentative D only over-estimates, so that is
// an upper bound of c* at any time; values above it cannot matter).  "Reached only by descending"
// is a property of the FIXPOINT (a tentative D(y) > V(y) may still come down to V(y)) and enters c*
// after the relaxation has converged.  Equal values among the
// touched cells are outside this equivalence (the heap's order of equal keys is its push order) -
// like the tie order of the level-set rule, parity-unpinned.
#include <climits>
#include "sl_common.h"

namespace {

constexpr double INF = __builtin_inf();

__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {
    // non-negative and negative doubles order differently as integers: compare-and-swap loop
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a;
    while (true) {
        const double cur = __longlong_as_double((long long)old);
        if (!(v < cur)) return;
        const unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
        if (prev == old) return;
        old = prev;
    }
}

struct RegionGrid { int d; int64_t n[SL_D]; int64_t nindex; };

__device__ __forceinline__ bool on_boundary(const RegionGrid& g, const int64_t* ijk) {
    bool b = false;
    for (int k = 0; k < g.d; ++k) b = b || ijk[k] == 0 || ijk[k] == g.n[k] - 1;
    return b;
}

__device__ __forceinline__ void unravel(const RegionGrid& g, int64_t idx, int64_t* ijk) {
    for (int k = g.d - 1; k >= 0; --k) { ijk[k] = idx % g.n[k]; idx /= g.n[k]; }
}

// visit the in-grid neighbours (3^d - 1 offsets) of ijk: f(flat index)
template <class F>
__device__ __forceinline__ void neighbours(const RegionGrid& g, const int64_t* ijk, F f) {
    int total = 1;
    for (int k = 0; k < g.d; ++k) total *= 3;
    for (int code = 0; code < total; ++code) {
        int c = code;
        int64_t flat = 0;
        bool inside = true, self = true;
        for (int k = 0; k < g.d; ++k) {            // most significant digit = first axis
            int div = 1;
            for (int j = k + 1; j < g.d; ++j) div *= 3;
            const int off = c / div - 1;
            c %= div;
            const int64_t v = ijk[k] + off;
            inside = inside && v >= 0 && v < g.n[k];
            self = self && off == 0;
            flat = flat * g.n[k] + v;
        }
        if (inside && !self) f(flat);
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_region_init(int64_t n, int64_t start,
                                                          const double* __restrict__ values,
                                                          double* __restrict__ dist, double* state) {
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SL_BLOCK)
        dist[i] = i == start ? values[i] : INF;
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = INF; state[1] = 0.0; }   // c_ub, changed
}

// one in-place relaxation pass; state[0] = best stop level known (prunes), state[1] = changed flag
__global__ __launch_bounds__(SL_BLOCK) void k_region_relax(RegionGrid g, int64_t start,
                                                           const double* __restrict__ values,
                                                           double* dist, double* state) {
    const double cub = state[0];
    bool changed = false;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < g.nindex;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        if (i == start) continue;
        int64_t ijk[SL_D];
        unravel(g, i, ijk);
        double m = INF;
        neighbours(g, ijk, [&](int64_t u) { const double du = dist[u]; m = du < m ? du : m; });
        if (!(m < INF)) continue;
        const double v = values[i];
        double cand = (v != v) ? INF : (v > m ? v : m);          // NaN: never reached
        if (cand < dist[i] && cand <= cub) { dist[i] = cand; changed = true; }
    }
    if (__any(changed) && (threadIdx.x & 63) == 0) state[1] = 1.0;
}

// state[0] = min(state[0], min of D over boundary nodes [and over nodes reached only by descending])
__global__ __launch_bounds__(SL_BLOCK) void k_region_bound(RegionGrid g, const double* __restrict__ values,
                                                           const double* __restrict__ dist, double* state,
                                                           bool with_descents) {
    double best = INF;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < g.nindex;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        const double di = dist[i];
        if (!(di < INF)) continue;
        int64_t ijk[SL_D];
        unravel(g, i, ijk);
        if (on_boundary(g, ijk) || (with_descents && di > values[i])) best = di < best ? di : best;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best < INF) atomic_min_f64(&state[0], best);
}

// The last regular pop x* (V = D = c*).  Several nodes can qualify (on a grid that is symmetric
// about the origin x and -x always tie): ONE of them is the reference's last pop - which one is
// decided by its heap's insertion counter, not pinned by anything - and the region holds that node
// and ITS descent neighbour only.  The lowest flat index is taken (pick), and the second kernel
// lets that node alone write out[0] = x*, out[1] = y (the lowest neighbour it descends to, -1 / -2).
__global__ __launch_bounds__(SL_BLOCK) void k_region_stop_pick(RegionGrid g, const double* __restrict__ values,
                                                               const double* __restrict__ dist,
                                                               const double* __restrict__ state,
                                                               long long* pick) {
    const double cstar = state[0];
    long long best = LLONG_MAX;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < g.nindex;
         i += (int64_t)gridDim.x * SL_BLOCK)
        if (dist[i] == cstar && values[i] == cstar) best = (long long)i < best ? (long long)i : best;
    for (int off = 32; off >= 1; off >>= 1) {
        const long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best != LLONG_MAX) atomicMin(pick, best);
}

__global__ __launch_bounds__(64) void k_region_stop(RegionGrid g, const double* __restrict__ values,
                                                    const double* __restrict__ dist,
                                                    const double* __restrict__ state,
                                                    const long long* __restrict__ pick, long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double cstar = state[0];
    const long long i = *pick;
    if (i == LLONG_MAX) return;                            // (out stays -1, -1)
    int64_t ijk[SL_D];
    unravel(g, i, ijk);
    long long y = -1;
    if (!on_boundary(g, ijk)) {
        double low = INF;
        neighbours(g, ijk, [&](int64_t u) {
            const double vu = values[u];
            if (vu < cstar && dist[u] == cstar && vu < low) { low = vu; y = (long long)u; }
        });
        if (y >= 0) {                                      // a descent onto the boundary is dropped
            int64_t yk[SL_D];
            unravel(g, y, yk);
            if (on_boundary(g, yk)) y = -2;
        }
        out[0] = i;
        out[1] = y;
    } else {
        out[0] = -1;                                       // x* on the boundary: dropped
        out[1] = -1;
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_region_mark(int64_t n, const double* __restrict__ dist,
                                                          const double* __restrict__ state,
                                                          const long long* __restrict__ stop,
                                                          uint8_t* __restrict__ region) {
    const double cstar = state[0];
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SL_BLOCK)
        region[i] = (dist[i] < cstar || i == stop[0] || i == stop[1]) ? 1 : 0;
}

}  // namespace

extern "C" int sl_lyapunov_region(sl_ctx* ctx, const double* d_values, int64_t start, double* d_work,
                                  uint8_t* d_region, int* sweeps_out) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_lyapunov_region: NULL context");
    if (!ctx->model_set) return sl_fail(ctx, SL_ERR_INVALID, "sl_lyapunov_region: call sl_model_set first");
    RegionGrid g;
    g.d = ctx->h_model.m.grid.d;
    g.nindex = ctx->h_model.gf.nindex;
    for (int k = 0; k < SL_D; ++k) g.n[k] = k < g.d ? ctx->h_model.m.grid.num_points[k] : 1;
    if (!d_values || !d_work || !d_region || start < 0 || start >= g.nindex)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_lyapunov_region: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // scratch: [c_ub, changed] + the two stop nodes + the picked last pop
    const size_t need = 5 * sizeof(double);
    if (need > ctx->scratch_bytes) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, 4096));
        ctx->scratch_bytes = 4096;
    }
    double* state = reinterpret_cast<double*>(ctx->d_scratch);
    long long* stop = reinterpret_cast<long long*>(state + 2);
    const int blocks = sl_grid_blocks(g.nindex);
    hipLaunchKernelGGL(k_region_init, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g.nindex, start,
                       d_values, d_work, state);
    int sweeps = 0;
    const int group = 8;                                   // passes between two looks at the flag
    int64_t limit = 4;
    for (int k = 0; k < g.d; ++k) limit += 2 * g.n[k];     // a path never needs more than ~sum n_k ... times
    limit *= 64;                                           // detours: generous bound, then give up loudly
    while (true) {
        double zero = 0.0;
        SL_HIP_CHECK(ctx, hipMemcpyAsync(state + 1, &zero, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        for (int r = 0; r < group; ++r) {
            hipLaunchKernelGGL(k_region_relax, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g, start,
                               d_values, d_work, state);
            hipLaunchKernelGGL(k_region_bound, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g, d_values,
                               d_work, state, false);
        }
        sweeps += group;
        double host[2];
        SL_HIP_CHECK(ctx, hipMemcpyAsync(host, state, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
        SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (host[1] == 0.0) break;
        if (sweeps > limit)
            return sl_fail(ctx, SL_ERR_HIP, "sl_lyapunov_region: no fixpoint after %d passes", sweeps);
    }
    hipLaunchKernelGGL(k_region_bound, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g, d_values, d_work,
                       state, true);                       // the fixpoint: descents count now
    long long none[3] = {-1, -1, LLONG_MAX};
    SL_HIP_CHECK(ctx, hipMemcpyAsync(stop, none, sizeof(none), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_region_stop_pick, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g, d_values, d_work,
                       state, stop + 2);
    hipLaunchKernelGGL(k_region_stop, dim3(1), dim3(64), 0, ctx->stream, g, d_values, d_work, state,
                       stop + 2, stop);
    hipLaunchKernelGGL(k_region_mark, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream, g.nindex, d_work, state,
                       stop, d_region);
    SL_HIP_CHECK(ctx, hipGetLastError());
    if (sweeps_out) *sweeps_out = sweeps;
    return SL_OK;
}
