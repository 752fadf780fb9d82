// sl_level.hip - the level-set rule of Lyapunov.update_safe_set (lyapunov.py:512-606) with every
// decision read from DEVICE memory, so that a whole update is one sequence of launches and
// collectives without a host round trip in between:
//
//   sl_lyap_sweep -> [gather of the 64-byte records] -> sl_fold_results
//                 -> sl_lyap_finalize_dev (key* = folded->fail, read by the kernel)
//                 -> [gather] -> sl_fold_results                      ... one D2H of 64 bytes
//   can_shrink = False / the no-failure c_max quirk add a radix select whose state (prefix, rank
//   among the remaining keys) lives on the device: sl_select_begin, then per byte sl_select_hist
//   -> [SUM all-reduce of 256 counters] -> sl_select_digit; the selected key feeds
//   sl_lyap_finalize_dev as key_keep.
//
// The ordering key is V on the grid points (lyapunov.py:305-322, 512).  For a QUADRATIC V the
// passes here do not read it from memory: a thread owns 8 consecutive cells of a row of the last
// grid axis and evaluates the ordered sums of functions.py:1534-1539 itself, sharing the prefix over
// the leading coordinates (same numbers, rounding included, as k_values writes: points by the
// np.linspace rule of functions.py:612-638).  The streaming pass of the 128^4 grid then moves 3 bits
// per cell instead of 8 bytes + 3 bits.
#include "sl_common.h"

namespace {

constexpr int CPT = SL_ROW_CELLS;        // cells per thread: one byte of every bit mask

// (the V of 8 consecutive cells and the fast key map live in sl_model.h - SlRowValues, sl_vbits_fast -
// so that the CPU suite compiles and checks the very same code)

template <int DT>
__device__ __forceinline__ void constants_to_vgprs(SlDevModel& M) {
    if (DT > 0) {
#pragma unroll
        for (int k = 0; k < (DT > 0 ? DT : 1); ++k) {
            asm volatile("" : "+v"(M.m.grid.unit_maxes[k]));
            asm volatile("" : "+v"(M.m.grid.offset[k]));
            asm volatile("" : "+v"(M.m.grid.upper[k]));
#pragma unroll
            for (int q = 0; q < (DT > 0 ? DT : 1); ++q) asm volatile("" : "+v"(M.m.value.matrix[k][q]));
        }
    }
}

__device__ __forceinline__ void reduce_finalize_block(
    const sl_key* partials, const int64_t* counts, int n, const sl_sweep_result* folded,
    sl_sweep_result* result, uint64_t* sv, int64_t* si, int64_t (*sc)[SL_BLOCK / 64]);

// ---- safe_i = init_i | key_i < key* | (prev_i & key_i >= key_keep) ------------------------------
// On small grids (`done` given) the workgroup that finishes LAST reduces the partials of all of them
// into the result record - one launch less per update, a fifth of such a step; on large ones the
// device-wide fence of 2048 workgroups costs more (0.016 ms at 128^4) than the launch it saves.
template <int DT>
__global__ __launch_bounds__(SL_BLOCK) void k_finalize_dev(
    const SlDevModel M_arg, int64_t lo, int64_t hi, const double* __restrict__ values,
    const uint8_t* __restrict__ init_bytes, const uint8_t* __restrict__ prev_bytes,
    const sl_sweep_result* __restrict__ folded, const sl_key* __restrict__ keep_ptr,
    uint8_t* __restrict__ safe_bytes, sl_key* partials, int64_t* counts,
    int span_groups, int vector_ok, unsigned int* done, sl_sweep_result* __restrict__ result) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ int64_t sc[2][SL_BLOCK / 64];
    SlDevModel M = M_arg;
    constants_to_vgprs<DT>(M);
    const sl_key star = folded->fail;
    sl_key keep;
    keep.vbits = ~0ull; keep.index = INT64_MAX;
    if (keep_ptr) keep = *keep_ptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ls_v = 0ull, mx_v = 0ull;
    int64_t ls_i = -1, mx_i = -1;
    int64_t n_below = 0, n_safe = 0;
    SlRowValues<DT> row;
    // Rows far above the level (nearly all of a large grid): one bound and one exact value per SPAN
    // of a row instead of every cell (SlRowValues::span_bounded; the launcher sets span_groups > 0 for
    // a quadratic V recomputed from the index, not negated, can_shrink = True - no previous set to
    // keep; with no failing key vstar is NaN and nothing is skipped)
    const bool bounded = DT > 0 && span_groups > 0;
    const double vstar = sl_vbits_to_double(star.vbits);
    const double margin = bounded ? SlRowValues<DT>::error_margin(M) : 0.0;
    // one group of eight cells (one byte of every mask) -> its safe bits; the statistics go to the
    // thread's running values.  Indices ascend over a thread's groups: ">=" on the value bits alone
    // keeps the lexicographic maxima.
    auto group = [&](int64_t i0, unsigned init8, unsigned prev8) -> unsigned {
        double v8[CPT];
        if (bounded && i0 + CPT <= hi) {
            const int top = row.eight_bounded(M, i0, vstar, margin, v8);
            if (top >= 0) {
                // every cell is above key*: the safe bits are the initial set's; the cell with the
                // largest value is the only candidate for the range's largest key
                const uint64_t vb = sl_vbits_fast(v8[top]);
                if (vb >= mx_v) { mx_v = vb; mx_i = i0 + top; }
                n_safe += __popc(init8);
                return init8;
            }
        } else {
            row.eight(M, values, lo, hi, i0, v8);
        }
        unsigned safe8 = 0u;
        auto one = [&](int c) {
            const int64_t idx = i0 + c;
            const uint64_t vb = sl_vbits_fast(v8[c]);
            const bool below = sl_key_less(vb, idx, star.vbits, star.index);
            const bool kept = ((prev8 >> c) & 1u) && !sl_key_less(vb, idx, keep.vbits, keep.index);
            const bool safe = below || kept || ((init8 >> c) & 1u);
            safe8 |= safe ? (1u << c) : 0u;
            if (below) { ++n_below; if (vb >= ls_v) { ls_v = vb; ls_i = idx; } }
            if (vb >= mx_v) { mx_v = vb; mx_i = idx; }
        };
        if (i0 + CPT <= hi) {                               // whole bytes: no per-cell range test
#pragma unroll
            for (int c = 0; c < CPT; ++c) one(c);
        } else {
            for (int c = 0; c < CPT; ++c) if (i0 + c < hi) one(c);
        }
        n_safe += __popc(safe8);
        return safe8;
    };
    if (bounded) {
        // One SPAN of span_groups bytes (a whole row of the last axis, or an aligned part of it: at
        // most 16 bytes) per thread and iteration: one bound for the span, and where it holds one
        // exact cell and a copy of the initial set's bytes.  Spans the bound does not clear fall
        // back to their groups of eight (bounded again one by one, then exact).
        const int span_cells = span_groups * CPT;
        const int64_t nthreads = (int64_t)gridDim.x * SL_BLOCK;
        for (int64_t s = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x;; s += nthreads) {
            const int64_t i0 = lo + s * span_cells;
            if (i0 >= hi) break;
            const int64_t byte0 = (i0 - lo) >> 3;
            if (i0 + span_cells <= hi) {
                double vt;
                const int top = row.span_bounded(M, i0, span_cells, vstar, margin, &vt);
                if (top >= 0) {
                    const uint64_t vb = sl_vbits_fast(vt);
                    if (vb >= mx_v) { mx_v = vb; mx_i = i0 + top; }
                    if (span_groups == 16 && vector_ok) {
                        uint4 w = make_uint4(0u, 0u, 0u, 0u);
                        if (init_bytes) w = *reinterpret_cast<const uint4*>(init_bytes + byte0);
                        n_safe += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
                        *reinterpret_cast<uint4*>(safe_bytes + byte0) = w;
                    } else {
                        for (int q = 0; q < span_groups; ++q) {
                            const unsigned init8 = init_bytes ? init_bytes[byte0 + q] : 0u;
                            n_safe += __popc(init8);
                            safe_bytes[byte0 + q] = (uint8_t)init8;
                        }
                    }
                    continue;
                }
            }
            for (int64_t j0 = i0; j0 < i0 + span_cells && j0 < hi; j0 += CPT)
                safe_bytes[(j0 - lo) >> 3] = (uint8_t)group(j0, init_bytes ? init_bytes[(j0 - lo) >> 3] : 0u, 0u);
        }
    } else {
        const int64_t span = (int64_t)SL_BLOCK * CPT;
        for (int64_t base = lo + (int64_t)blockIdx.x * span; base < hi; base += (int64_t)gridDim.x * span) {
            const int64_t i0 = base + (int64_t)threadIdx.x * CPT;
            if (i0 >= hi) continue;
            const unsigned init8 = init_bytes ? init_bytes[(i0 - lo) >> 3] : 0u;
            const unsigned prev8 = prev_bytes ? prev_bytes[(i0 - lo) >> 3] : 0u;
            safe_bytes[(i0 - lo) >> 3] = (uint8_t)group(i0, init8, prev8);
        }
    }
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
    // last workgroup done: its partials are visible device-wide before the count goes up, the
    // reader's fence makes it see everybody else's
    __shared__ int last;
    if (!done) return;                       // (large grids: k_reduce_finalize_dev follows)
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        reduce_finalize_block(partials, counts, (int)gridDim.x, folded, result, sv, si, sc);
        if (threadIdx.x == 0) *done = 0u;
    }
}

// partials of k_finalize_dev -> last_safe, max_key, count_below, count_safe of the result record;
// `fail` is carried over from the folded record the pass was given (a later fold keeps it).  One
// workgroup: the last one of k_finalize_dev to finish, or k_reduce_finalize_dev (empty ranges).
__device__ __forceinline__ void reduce_finalize_block(
    const sl_key* partials, const int64_t* counts, int n, const sl_sweep_result* folded,
    sl_sweep_result* result, uint64_t* sv, int64_t* si, int64_t (*sc)[SL_BLOCK / 64]) {
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
    __syncthreads();                                   // (the callers' use of the scratch is over)
    if (lane == 0) { sc[0][wave] = a; sc[1][wave] = b; }
    sl_block_reduce_key<false>(ls_v, ls_i, sv, si);
    __syncthreads();
    sl_block_reduce_key<false>(mx_v, mx_i, sv, si);
    if (threadIdx.x == 0) {
        const sl_key fail = folded->fail;
        result->fail = fail;
        result->last_safe.vbits = ls_v; result->last_safe.index = ls_i;
        result->max_key.vbits = mx_v; result->max_key.index = mx_i;
        int64_t ta = 0, tb = 0;
        for (int w = 0; w < SL_BLOCK / 64; ++w) { ta += sc[0][w]; tb += sc[1][w]; }
        result->count_below = ta; result->count_safe = tb;
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_reduce_finalize_dev(
    const sl_key* __restrict__ partials, const int64_t* __restrict__ counts, int n,
    const sl_sweep_result* __restrict__ folded, sl_sweep_result* __restrict__ result) {
    __shared__ uint64_t sv[SL_BLOCK / 64];
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ int64_t sc[2][SL_BLOCK / 64];
    reduce_finalize_block(partials, counts, n, folded, result, sv, si, sc);
}

// Refinement N(x) after a NON-adaptive update that must not shrink (lyapunov.py:507-510, 531,
// 585-587, 601-606): the loop writes 1 where the decrease condition holds, 0 from the first failure
// to the end of its batch, and leaves every other cell what an earlier ADAPTIVE update made it -
//   ref_i <- init_i ? 1 : key_i < key* ? (negative_i ? 1 : ref_i) : (key_i >= key_keep ? ref_i : 0).
template <int DT>
__global__ __launch_bounds__(SL_BLOCK) void k_refinement_carry(
    const SlDevModel M_arg, int64_t lo, int64_t hi, const double* __restrict__ values,
    const uint8_t* __restrict__ init_bytes, const uint8_t* __restrict__ neg_bytes,
    const sl_sweep_result* __restrict__ folded, const sl_key* __restrict__ keep_ptr,
    int64_t* __restrict__ refinement) {
    SlDevModel M = M_arg;
    constants_to_vgprs<DT>(M);
    const sl_key star = folded->fail;
    sl_key keep;
    keep.vbits = ~0ull; keep.index = INT64_MAX;
    if (keep_ptr) keep = *keep_ptr;
    SlRowValues<DT> row;
    const int64_t span = (int64_t)SL_BLOCK * CPT;
    for (int64_t base = lo + (int64_t)blockIdx.x * span; base < hi; base += (int64_t)gridDim.x * span) {
        const int64_t i0 = base + (int64_t)threadIdx.x * CPT;
        if (i0 >= hi) continue;
        const unsigned init8 = init_bytes ? init_bytes[(i0 - lo) >> 3] : 0u;
        const unsigned neg8 = neg_bytes[(i0 - lo) >> 3];
        double v8[CPT];
        row.eight(M, values, lo, hi, i0, v8);
        for (int c = 0; c < CPT; ++c) {
            const int64_t idx = i0 + c;
            if (idx >= hi) break;
            const uint64_t vb = sl_vbits_fast(v8[c]);
            const bool below = sl_key_less(vb, idx, star.vbits, star.index);
            const bool later = !sl_key_less(vb, idx, keep.vbits, keep.index);
            const int64_t old = refinement[idx - lo];
            int64_t now = below ? (((neg8 >> c) & 1u) ? 1 : old) : (later ? old : 0);
            if ((init8 >> c) & 1u) now = 1;
            refinement[idx - lo] = now;
        }
    }
}

// records[count] -> out: the reductions of lyapunov.py:512-606 over the shards
__global__ void k_fold_records(const sl_sweep_result* __restrict__ records, int count,
                               sl_sweep_result* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sl_sweep_result r = records[0];
    for (int k = 1; k < count; ++k) {
        const sl_sweep_result o = records[k];
        if (sl_key_less(o.fail.vbits, o.fail.index, r.fail.vbits, r.fail.index)) r.fail = o.fail;
        if (sl_key_less(r.last_safe.vbits, r.last_safe.index, o.last_safe.vbits, o.last_safe.index))
            r.last_safe = o.last_safe;
        if (sl_key_less(r.max_key.vbits, r.max_key.index, o.max_key.vbits, o.max_key.index))
            r.max_key = o.max_key;
        r.count_below += o.count_below;
        r.count_safe += o.count_safe;
    }
    *out = r;
}

// ---- radix select of the k-th smallest (V, index) key, state on the device ----------------------
__global__ void k_select_begin(sl_select_state* state, int64_t k, int64_t batch,
                               const sl_sweep_result* __restrict__ folded, int64_t n_total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sl_select_state s;
    // k < 0: the first position of the batch AFTER the one that holds the first failure
    // (lyapunov.py:585-587 never touches later batches): (#keys below key* / batch + 1) * batch
    int64_t rank = k;
    if (k < 0) rank = (folded->count_below / batch + 1) * batch;
    s.rank = rank;
    s.none = (rank < 0 || rank >= n_total) ? 1 : 0;
    s.prefix = 0ull;
    s.remaining = rank;
    s.key.vbits = ~0ull;
    s.key.index = INT64_MAX;
    s.pad[0] = s.pad[1] = 0;
    *state = s;
}

template <int DT>
__global__ __launch_bounds__(SL_BLOCK) void k_select_hist(
    const SlDevModel M_arg, int64_t lo, int64_t hi, const double* __restrict__ values, int which,
    int byte, const sl_select_state* __restrict__ state, uint64_t* __restrict__ hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    if (state->none) return;                             // uniform: every thread leaves
    SlDevModel M = M_arg;
    constants_to_vgprs<DT>(M);
    const uint64_t prefix = state->prefix, vbits_equal = state->key.vbits;
    const int shift = byte * 8;
    const uint64_t himask = (byte == 7) ? 0ull : (~0ull << (shift + 8));
    SlRowValues<DT> row;
    const int64_t span = (int64_t)SL_BLOCK * CPT;
    for (int64_t base = lo + (int64_t)blockIdx.x * span; base < hi; base += (int64_t)gridDim.x * span) {
        const int64_t i0 = base + (int64_t)threadIdx.x * CPT;
        if (i0 >= hi) continue;
        double v8[CPT];
        row.eight(M, values, lo, hi, i0, v8);
        auto one = [&](int c) {
            const int64_t idx = i0 + c;
            const uint64_t vb = sl_vbits_fast(v8[c]);
            uint64_t key;
            bool take;
            if (which == 0) { key = vb; take = true; }
            else { key = (uint64_t)idx; take = (vb == vbits_equal); }
            take = take && ((key & himask) == (prefix & himask));
            if (take) atomicAdd(&lh[(key >> shift) & 0xff], 1u);
        };
        if (i0 + CPT <= hi) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) one(c);
        } else {
            for (int c = 0; c < CPT; ++c) if (i0 + c < hi) one(c);
        }
    }
    __syncthreads();
    const unsigned int cnt = lh[threadIdx.x];
    if (cnt) atomicAdd((unsigned long long*)&hist[threadIdx.x], (unsigned long long)cnt);
}

// 256 (all-reduced) counters -> the digit that holds the wanted rank; the state moves on
__global__ __launch_bounds__(256) void k_select_digit(int which, int byte,
                                                      const uint64_t* __restrict__ hist,
                                                      sl_select_state* state) {
    __shared__ uint64_t cum[256];
    cum[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        sl_select_state s = *state;
        if (!s.none) {
            uint64_t before = 0ull;
            int digit = 255;
            for (int b = 0; b < 256; ++b) {
                if (before + cum[b] > (uint64_t)s.remaining) { digit = b; break; }
                before += cum[b];
            }
            s.remaining -= (int64_t)before;
            s.prefix |= (uint64_t)digit << (8 * byte);
            if (byte == 0) {
                if (which == 0) { s.key.vbits = s.prefix; s.prefix = 0ull; }   // now: index among equal values
                else s.key.index = (int64_t)s.prefix;
            }
            *state = s;
        }
    }
}

int dim_variant(sl_ctx* ctx, const double* d_values, const char* who, int* dt) {
    *dt = 0;
    if (d_values) return SL_OK;
    int ok = 0;
    sl_values_implicit(ctx, &ok);
    if (!ok)
        return sl_fail(ctx, SL_ERR_INVALID, "%s: NULL values need a quadratic V on a grid of at most "
                                            "4 dimensions (sl_values_implicit)", who);
    *dt = ctx->h_model.m.grid.d;
    return SL_OK;
}

int blocks_for(sl_ctx* ctx, int64_t cells) {
    const int64_t span = (int64_t)SL_BLOCK * CPT;
    int64_t blocks = (cells + span - 1) / span;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks > SL_MAX_GRID) blocks = SL_MAX_GRID;
    return (int)blocks;
}

}  // namespace

// May the passes of this file (and the sweep) be called with d_values = NULL?  Yes for a quadratic
// V on a grid of 1..4 dimensions with a last axis of whole bytes (a multiple of 8 cells) whose
// points by the np.linspace rule (last point = the upper limit) equal index_to_state bit for bit -
// then the sweep's own V(x) IS the ordering key.
extern "C" int sl_values_implicit(sl_ctx* ctx, int* out) {
    if (!ctx || !out) return sl_fail(ctx, SL_ERR_INVALID, "sl_values_implicit: NULL argument");
    *out = 0;
    if (!ctx->model_set) return SL_OK;
    if (!sl_values_implicit_ok(ctx->h_model.m)) return SL_OK;
    *out = 1;
    return SL_OK;
}

extern "C" int sl_fold_results(sl_ctx* ctx, const sl_sweep_result* d_records, int count,
                               sl_sweep_result* d_out) {
    if (!ctx || !d_records || !d_out || count < 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_fold_results: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fold_records, dim3(1), dim3(64), 0, ctx->stream, d_records, count, d_out);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_lyap_finalize_dev(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                                    const uint64_t* d_init_bits, const uint64_t* d_prev_bits,
                                    const sl_sweep_result* d_folded, const sl_key* d_keep,
                                    uint64_t* d_safe_bits, sl_sweep_result* d_result) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_lyap_finalize_dev: NULL context");
    if (lo < 0 || hi < lo || ((lo & 63) && hi != lo) || !d_folded || !d_safe_bits || !d_result)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_lyap_finalize_dev: bad argument");
    int dt = 0;
    int rc = dim_variant(ctx, d_values, "sl_lyap_finalize_dev", &dt);
    if (rc) return rc;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SlTimed timed(ctx, 1);
    int blocks = (hi == lo) ? 0 : blocks_for(ctx, hi - lo);
    // The bounded pass (SlRowValues::span_bounded): a quadratic V recomputed from the index, not
    // negated, can_shrink = True (no previous set to keep).  A thread's span: the largest part of a
    // row of the last axis, at most 16 bytes of the masks, that the range starts on a multiple of.
    int span_groups = 0, vector_ok = 0;
    if (blocks && dt > 0 && !d_prev_bits && !ctx->h_model.m.value.negate) {
        const int64_t row_groups = ctx->h_model.m.grid.num_points[dt - 1] / CPT;
        // (long spans only where they leave the device enough threads: a 256 x 256 grid is 8192
        // groups of eight - one thread each, as in the unbounded pass)
        for (int g = 16; g >= 1 && !span_groups; --g)
            if (row_groups % g == 0 && lo % ((int64_t)g * CPT) == 0 &&
                (g == 1 || (hi - lo) / ((int64_t)g * CPT) >= 65536)) span_groups = g;
        vector_ok = (reinterpret_cast<uintptr_t>(d_safe_bits) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(d_init_bits) % 16 == 0);
        const int64_t spans = (hi - lo + (int64_t)span_groups * CPT - 1) / ((int64_t)span_groups * CPT);
        const int64_t need = (spans + SL_BLOCK - 1) / SL_BLOCK;
        if (need < blocks) blocks = (int)need;
    }
    const bool fold_tail = blocks <= 256;
    if (blocks) {
        if ((hi - lo) & 63)     // whole bytes are written: clear the rest of the last mask word first
            SL_HIP_CHECK(ctx, hipMemsetAsync(d_safe_bits + ((hi - lo) >> 6), 0, sizeof(uint64_t), ctx->stream));
        const uint8_t* init_bytes = reinterpret_cast<const uint8_t*>(d_init_bits);
        const uint8_t* prev_bytes = reinterpret_cast<const uint8_t*>(d_prev_bits);
        uint8_t* safe_bytes = reinterpret_cast<uint8_t*>(d_safe_bits);
#define SL_FIN(D_)                                                                                  \
    hipLaunchKernelGGL(k_finalize_dev<D_>, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream,            \
                       ctx->h_model, lo, hi, d_values, init_bytes, prev_bytes, d_folded, d_keep,    \
                       safe_bytes, ctx->d_partials, ctx->d_partial_counts, span_groups, vector_ok, \
                       fold_tail ? reinterpret_cast<unsigned int*>(ctx->d_ticket + 1) : nullptr, d_result)
        switch (dt) {
            case 1: SL_FIN(1); break;
            case 2: SL_FIN(2); break;
            case 3: SL_FIN(3); break;
            case 4: SL_FIN(4); break;
            default: SL_FIN(0); break;
        }
#undef SL_FIN
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    if (!blocks || !fold_tail) {                        // (an empty range: the record of no cells)
        hipLaunchKernelGGL(k_reduce_finalize_dev, dim3(1), dim3(SL_BLOCK), 0, ctx->stream,
                           ctx->d_partials, ctx->d_partial_counts, blocks, d_folded, d_result);
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    return SL_OK;
}

extern "C" int sl_refinement_carry(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                                  const uint64_t* d_init_bits, const uint64_t* d_neg_bits,
                                  const sl_sweep_result* d_folded, const sl_key* d_keep,
                                  int64_t* d_refinement) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_refinement_carry: NULL context");
    if (lo < 0 || hi < lo || ((lo & 63) && hi != lo) || !d_folded || !d_neg_bits || !d_refinement)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_refinement_carry: bad argument");
    int dt = 0;
    int rc = dim_variant(ctx, d_values, "sl_refinement_carry", &dt);
    if (rc) return rc;
    if (hi == lo) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int blocks = blocks_for(ctx, hi - lo);
    const uint8_t* init_bytes = reinterpret_cast<const uint8_t*>(d_init_bits);
    const uint8_t* neg_bytes = reinterpret_cast<const uint8_t*>(d_neg_bits);
#define SL_CARRY(D_)                                                                                \
    hipLaunchKernelGGL(k_refinement_carry<D_>, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream,        \
                       ctx->h_model, lo, hi, d_values, init_bytes, neg_bytes, d_folded, d_keep,     \
                       d_refinement)
    switch (dt) {
        case 1: SL_CARRY(1); break;
        case 2: SL_CARRY(2); break;
        case 3: SL_CARRY(3); break;
        case 4: SL_CARRY(4); break;
        default: SL_CARRY(0); break;
    }
#undef SL_CARRY
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_select_begin(sl_ctx* ctx, sl_select_state* d_state, int64_t k, int64_t batch,
                               const sl_sweep_result* d_folded, int64_t n_total) {
    if (!ctx || !d_state || n_total < 0 || (k < 0 && (!d_folded || batch < 1)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_select_begin: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_select_begin, dim3(1), dim3(64), 0, ctx->stream, d_state, k, batch, d_folded,
                       n_total);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_select_hist(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values, int which,
                              int byte, const sl_select_state* d_state, uint64_t* d_hist) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "sl_select_hist: NULL context");
    if (lo < 0 || hi < lo || !d_state || !d_hist || byte < 0 || byte > 7 || which < 0 || which > 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_select_hist: bad argument");
    int dt = 0;
    int rc = dim_variant(ctx, d_values, "sl_select_hist", &dt);
    if (rc) return rc;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
    if (hi == lo) return SL_OK;
    const int blocks = blocks_for(ctx, hi - lo);
#define SL_HIST(D_)                                                                                 \
    hipLaunchKernelGGL(k_select_hist<D_>, dim3(blocks), dim3(SL_BLOCK), 0, ctx->stream,             \
                       ctx->h_model, lo, hi, d_values, which, byte, d_state, d_hist)
    switch (dt) {
        case 1: SL_HIST(1); break;
        case 2: SL_HIST(2); break;
        case 3: SL_HIST(3); break;
        case 4: SL_HIST(4); break;
        default: SL_HIST(0); break;
    }
#undef SL_HIST
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_select_digit(sl_ctx* ctx, int which, int byte, const uint64_t* d_hist,
                               sl_select_state* d_state) {
    if (!ctx || !d_hist || !d_state || byte < 0 || byte > 7 || which < 0 || which > 1)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_select_digit: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_select_digit, dim3(1), dim3(256), 0, ctx->stream, which, byte, d_hist, d_state);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}
