// sl_sample.hip - the glue of get_safe_sample / perturb_actions (lyapunov.py:609-797) as kernels:
// states of the safe cells, the perturbed and clipped state-action candidates, their duplicate
// removal in the byte-wise order of utilities.unique_rows (:496-516), the confidence bound and the
// level-set test of a candidate, membership of its successor in the safe set and the arg-max.
// (The GP posterior, V and L_v at the candidates come from sl_eval_points; the safe cells are
// compacted with sl_partition_by_digit, rows are ordered with sl_sort_pairs.)
#include "sl_common.h"

namespace {

__global__ __launch_bounds__(SL_BLOCK) void k_index_to_state(const SlDevModel M, int64_t count,
                                                             const int64_t* __restrict__ indices,
                                                             double* __restrict__ states) {
    const int d = M.m.grid.d;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        double x[SL_P];
        sl_index_to_state(M.m.grid, M.gf, d, indices[i], x);           // functions.py:714-731
        for (int k = 0; k < d; ++k) states[i * d + k] = x[k];
    }
}

// row (i, k) = [state_i, clip(action_i + perturbation_k)]                          lyapunov.py:634-647
__global__ __launch_bounds__(SL_BLOCK) void k_perturb_pairs(
    int64_t count, int d, int m, const double* __restrict__ states, const double* __restrict__ actions,
    int nperturb, const double* __restrict__ perturbations, const double* __restrict__ limits,
    double* __restrict__ pairs) {
    const int64_t total = count * nperturb;
    const int w = d + m;
    for (int64_t r = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; r < total;
         r += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t i = r / nperturb;
        const int k = (int)(r - i * nperturb);
        for (int c = 0; c < d; ++c) pairs[r * w + c] = states[i * d + c];
        for (int a = 0; a < m; ++a) {
            double u = actions[i * m + a] + perturbations[k * m + a];
            if (limits) {                                                      // np.clip: min(max(u, lo), hi)
                const double lo = limits[2 * a], hi = limits[2 * a + 1];
                u = u < lo ? lo : u;
                u = u > hi ? hi : u;
            }
            pairs[r * w + d + a] = u;
        }
    }
}

// memcmp order of a little-endian float64 = unsigned order of its byte-swapped bits
__global__ __launch_bounds__(SL_BLOCK) void k_rows_sort_key(int64_t count, int words, int column,
                                                            const int64_t* __restrict__ rows,
                                                            const int64_t* __restrict__ order,
                                                            uint64_t* __restrict__ keys) {
    for (int64_t q = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; q < count;
         q += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t r = order ? order[q] : q;
        keys[q] = __builtin_bswap64((uint64_t)rows[r * words + column]);
    }
}

// flags[q] = 1 when the q-th row (in `order`) repeats the one before it, bit for bit
__global__ __launch_bounds__(SL_BLOCK) void k_rows_duplicate_flags(int64_t count, int words,
                                                                   const int64_t* __restrict__ rows,
                                                                   const int64_t* __restrict__ order,
                                                                   uint8_t* __restrict__ flags) {
    for (int64_t q = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; q < count;
         q += (int64_t)gridDim.x * SL_BLOCK) {
        bool same = q > 0;
        if (same) {
            const int64_t a = order[q], b = order[q - 1];
            for (int w = 0; w < words; ++w) same = same && rows[a * words + w] == rows[b * words + w];
        }
        flags[q] = same ? 1 : 0;
    }
}

// bound = sum_j std_j; inside = V(mean) + sum_j L_v(mean)_j std_j < c_max               lyapunov.py:716-726
__global__ __launch_bounds__(SL_BLOCK) void k_sample_bounds(
    int64_t count, int d, int lv_cols, const double* __restrict__ std, const double* __restrict__ lv,
    const double* __restrict__ value, double c_max, double* __restrict__ bound,
    uint8_t* __restrict__ inside) {
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        double b = std[i * d], e = lv[i * lv_cols] * std[i * d];
        for (int k = 1; k < d; ++k) {                      // left to right like the oracle
            b = b + std[i * d + k];
            const double t = lv[i * lv_cols + (lv_cols > 1 ? k : 0)] * std[i * d + k];
            e = e + t;
        }
        bound[i] = b;
        inside[i] = (value[i] + e) < c_max ? 1 : 0;
    }
}

// out[i] &= safe_set[state_to_index(point_i)]                          lyapunov.py:762-766, functions.py:733-752
__global__ __launch_bounds__(SL_BLOCK) void k_state_membership(const SlDevModel M, int64_t count,
                                                               const double* __restrict__ points,
                                                               const uint64_t* __restrict__ safe_bits,
                                                               uint8_t* __restrict__ inout) {
    const int d = M.m.grid.d;
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        int64_t flat = 0;
        bool nan = false;
        for (int k = 0; k < d; ++k) {
            double x = points[i * d + k];
            nan = nan || (x != x);             // a NaN posterior maps nowhere (the reference raises in
                                               // ravel_multi_index): not inside, and no look-up
            const double lo = M.m.grid.offset[k], hi = M.m.grid.upper[k];
            x = x < lo ? lo : x;                                               // np.clip
            x = x > hi ? hi : x;
            const double inv = 1.0 / M.m.grid.unit_maxes[k];
            const double s = (x - lo) * inv;
            int64_t ijk = nan ? 0 : (int64_t)rint(s);                          // np.rint: half to even
            const int64_t last = M.m.grid.num_points[k] - 1;
            ijk = ijk < 0 ? 0 : (ijk > last ? last : ijk);                     // (rounding at the upper limit)
            flat = flat * M.m.grid.num_points[k] + ijk;
        }
        const bool safe = !nan && ((safe_bits[flat >> 6] >> (flat & 63)) & 1ull);
        inout[i] = (inout[i] && safe) ? 1 : 0;
    }
}

// first index of the largest value among the rows with mask != 0 (np.argmax: NaN counts as largest)
__global__ __launch_bounds__(SL_BLOCK) void k_argmax_masked(int64_t count, const double* __restrict__ values,
                                                            const uint8_t* __restrict__ mask,
                                                            int64_t* __restrict__ out) {
    __shared__ double sv[SL_BLOCK];
    __shared__ int64_t si[SL_BLOCK];
    __shared__ int64_t sn[SL_BLOCK];
    double best = 0.0;
    int64_t at = -1, n = 0;
    auto better = [](double a, int64_t ia, double b, int64_t ib) {          // (a, ia) beats (b, ib)
        if (ib < 0) return true;
        const bool an = a != a, bn = b != b;
        if (an != bn) return an;
        if (!an && a != b) return a > b;
        return ia < ib;
    };
    for (int64_t i = threadIdx.x; i < count; i += SL_BLOCK) {
        if (mask && !mask[i]) continue;
        ++n;
        if (better(values[i], i, best, at)) { best = values[i]; at = i; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = at; sn[threadIdx.x] = n;
    __syncthreads();
    for (int off = SL_BLOCK / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            const int o = threadIdx.x + off;
            if (si[o] >= 0 && better(sv[o], si[o], sv[threadIdx.x], si[threadIdx.x])) {
                sv[threadIdx.x] = sv[o]; si[threadIdx.x] = si[o];
            }
            sn[threadIdx.x] += sn[o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = si[0]; out[1] = sn[0]; }
}

int check_model(sl_ctx* ctx, const char* who) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "%s: NULL context", who);
    if (!ctx->model_set) return sl_fail(ctx, SL_ERR_INVALID, "%s: call sl_model_set first", who);
    return SL_OK;
}


// ---- np.where(safe_set) (lyapunov.py:729): the flat indices of the set bits, ascending ------------------
// A workgroup owns 256 consecutive mask words (16 384 cells).  k_bits_count: set bits per workgroup;
// k_bits_scan (one workgroup): exclusive scan of those counts in place, the total behind them;
// k_bits_scatter: a thread's word starts at (workgroup offset + bits of the words in front of it
// in the workgroup) and writes the positions of its set bits in ascending order.  Reads the mask
// (n / 8 bytes) and writes 8 bytes per set bit - the stable partition of all n cells it replaces
// wrote 8 n bytes (2.1 GB at 128^4) to find them.
__device__ __forceinline__ uint64_t bits_word(int64_t n, const uint64_t* __restrict__ bits, int64_t w) {
    const int64_t nwords = (n + 63) >> 6;
    if (w >= nwords) return 0ull;
    uint64_t v = bits[w];
    const int tail = (int)(n & 63);
    if (w == nwords - 1 && tail) v &= (1ull << tail) - 1ull;      // bits beyond n are not cells
    return v;
}

__global__ __launch_bounds__(256) void k_bits_count(int64_t n, const uint64_t* __restrict__ bits,
                                                    uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t part[4];
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t c = (uint32_t)__popcll(bits_word(n, bits, w));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// counts [nblocks] -> exclusive prefix sums (64-bit: 2^32 set bits and more at 128^4 x 16) in
// offsets [nblocks + 1], offsets[nblocks] = total
__global__ __launch_bounds__(1024) void k_bits_scan(int64_t nblocks, const uint32_t* __restrict__ counts,
                                                    int64_t* __restrict__ offsets) {
    __shared__ int64_t chunk_sum[1024];
    const int t = threadIdx.x;
    const int64_t per = (nblocks + 1023) / 1024, b0 = (int64_t)t * per;
    const int64_t b1 = b0 + per < nblocks ? b0 + per : nblocks;
    int64_t sum = 0;
    for (int64_t b = b0; b < b1; ++b) sum += counts[b];
    chunk_sum[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 1024 chunk sums
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = t >= off ? chunk_sum[t - off] : 0;
        __syncthreads();
        chunk_sum[t] += add;
        __syncthreads();
    }
    int64_t run = t ? chunk_sum[t - 1] : 0;
    for (int64_t b = b0; b < b1; ++b) {
        offsets[b] = run;
        run += counts[b];
    }
    if (t == 1023) offsets[nblocks] = chunk_sum[1023];
}

__global__ __launch_bounds__(256) void k_bits_scatter(int64_t n, const uint64_t* __restrict__ bits,
                                                      const int64_t* __restrict__ offsets,
                                                      int64_t* __restrict__ indices) {
    __shared__ uint32_t wave_sum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t v = bits_word(n, bits, w);
    const uint32_t c = (uint32_t)__popcll(v);
    // exclusive scan of the popcounts over the wavefront, then over the four wavefronts
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (int k = 0; k < wave; ++k) before += wave_sum[k];
    int64_t at = offsets[blockIdx.x] + before;
    const int64_t base = w << 6;
    while (v) {
        indices[at++] = base + __builtin_ctzll(v);
        v &= v - 1;
    }
}
}  // namespace

extern "C" int sl_index_to_state(sl_ctx* ctx, int64_t count, const int64_t* d_indices, double* d_states) {
    int rc = check_model(ctx, "sl_index_to_state");
    if (rc) return rc;
    if (count < 0 || (count && (!d_indices || !d_states)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_index_to_state: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_index_to_state, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream,
                       ctx->h_model, count, d_indices, d_states);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_perturb_pairs(sl_ctx* ctx, int64_t count, int d, int m, const double* d_states,
                                const double* d_actions, int nperturb, const double* d_perturbations,
                                const double* d_limits, double* d_pairs) {
    if (!ctx || count < 0 || d < 1 || m < 1 || nperturb < 1 || !d_perturbations ||
        (count && (!d_states || !d_actions || !d_pairs)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_perturb_pairs: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_perturb_pairs, dim3(sl_grid_blocks(count * nperturb)), dim3(SL_BLOCK), 0,
                       ctx->stream, count, d, m, d_states, d_actions, nperturb, d_perturbations, d_limits,
                       d_pairs);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_rows_sort_key(sl_ctx* ctx, int64_t count, int words, int column, const int64_t* d_rows,
                                const int64_t* d_order, uint64_t* d_keys) {
    if (!ctx || count < 0 || words < 1 || column < 0 || column >= words || (count && (!d_rows || !d_keys)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_rows_sort_key: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_rows_sort_key, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream, count,
                       words, column, d_rows, d_order, d_keys);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_rows_duplicate_flags(sl_ctx* ctx, int64_t count, int words, const int64_t* d_rows,
                                       const int64_t* d_order, uint8_t* d_flags) {
    if (!ctx || count < 0 || words < 1 || (count && (!d_rows || !d_order || !d_flags)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_rows_duplicate_flags: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_rows_duplicate_flags, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream,
                       count, words, d_rows, d_order, d_flags);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_sample_bounds(sl_ctx* ctx, int64_t count, int d, int lv_cols, const double* d_std,
                                const double* d_lv, const double* d_value, double c_max, double* d_bound,
                                uint8_t* d_inside) {
    if (!ctx || count < 0 || d < 1 || (lv_cols != 1 && lv_cols != d) ||
        (count && (!d_std || !d_lv || !d_value || !d_bound || !d_inside)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_sample_bounds: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sample_bounds, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream, count,
                       d, lv_cols, d_std, d_lv, d_value, c_max, d_bound, d_inside);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_state_membership(sl_ctx* ctx, int64_t count, const double* d_points,
                                   const uint64_t* d_safe_bits, uint8_t* d_inout) {
    int rc = check_model(ctx, "sl_state_membership");
    if (rc) return rc;
    if (count < 0 || !d_safe_bits || (count && (!d_points || !d_inout)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_state_membership: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_state_membership, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream,
                       ctx->h_model, count, d_points, d_safe_bits, d_inout);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// best[i] = np.argmax over the actions of q[i][:] with the actions that are not allowed at vertex i
// set to -inf (reinforcement_learning.py:266-278): the first index wins, a NaN counts as largest,
// all actions ruled out -> index 0.  allowed[a] is a bit mask over the vertices (bit i of word i >> 6).
__global__ __launch_bounds__(SL_BLOCK) void k_argmax_rows_masked(int64_t count, int n_actions,
                                                                 const double* __restrict__ q,
                                                                 const uint64_t* __restrict__ allowed,
                                                                 int64_t words, int32_t* __restrict__ best) {
    const double NEG_INF = -__builtin_inf();
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        double top = 0.0;
        int at = -1;
        for (int a = 0; a < n_actions; ++a) {
            double v = q[i * n_actions + a];
            if (allowed && !((allowed[(int64_t)a * words + (i >> 6)] >> (i & 63)) & 1ull)) v = NEG_INF;
            const bool vn = v != v, tn = top != top;
            if (at < 0 || (vn && !tn) || (!vn && !tn && v > top)) { top = v; at = a; }
        }
        best[i] = at < 0 ? 0 : at;
    }
}

extern "C" int sl_argmax_rows_masked(sl_ctx* ctx, int64_t count, int n_actions, const double* d_q,
                                     const uint64_t* d_allowed_bits, int64_t words_per_action,
                                     int32_t* d_best) {
    if (!ctx || count < 0 || n_actions < 1 || !d_best || (count && !d_q) ||
        (d_allowed_bits && words_per_action * 64 < count))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_argmax_rows_masked: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (count) {
        hipLaunchKernelGGL(k_argmax_rows_masked, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream,
                           count, n_actions, d_q, d_allowed_bits, words_per_action, d_best);
        SL_HIP_CHECK(ctx, hipGetLastError());
    }
    return SL_OK;
}

extern "C" int sl_argmax_masked(sl_ctx* ctx, int64_t count, const double* d_values, const uint8_t* d_mask,
                                int64_t* d_out) {
    if (!ctx || count < 0 || !d_out || (count && !d_values))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_argmax_masked: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_argmax_masked, dim3(1), dim3(SL_BLOCK), 0, ctx->stream, count, d_values, d_mask,
                       d_out);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_bits_count(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, uint32_t* d_block_counts,
                             int64_t* d_offsets, int64_t* total_out) {
    if (!ctx || n < 0 || !total_out || (n && (!d_bits || !d_block_counts || !d_offsets)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bits_count: bad argument");
    *total_out = 0;
    if (n == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t nwords = (n + 63) >> 6, nblocks = (nwords + 255) / 256;
    hipLaunchKernelGGL(k_bits_count, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, n, d_bits, d_block_counts);
    hipLaunchKernelGGL(k_bits_scan, dim3(1), dim3(1024), 0, ctx->stream, nblocks, d_block_counts, d_offsets);
    SL_HIP_CHECK(ctx, hipGetLastError());
    SL_HIP_CHECK(ctx, hipMemcpyAsync(total_out, d_offsets + nblocks, sizeof(int64_t), hipMemcpyDeviceToHost,
                                     ctx->stream));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SL_OK;
}

extern "C" int sl_bits_to_indices(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, const int64_t* d_offsets,
                                  int64_t* d_indices) {
    if (!ctx || n < 0 || (n && (!d_bits || !d_offsets || !d_indices)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_bits_to_indices: bad argument");
    if (n == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t nwords = (n + 63) >> 6, nblocks = (nwords + 255) / 256;
    hipLaunchKernelGGL(k_bits_scatter, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, n, d_bits, d_offsets,
                       d_indices);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}
