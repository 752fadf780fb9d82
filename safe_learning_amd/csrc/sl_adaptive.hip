// sl_adaptive.hip - the adaptive branch of Lyapunov.update_safe_set (lyapunov.py:445-487, 540-582)
// and the device primitives it needs: a stable LSD radix sort of (vbits, payload) pairs, a stable
// partition by an 8-bit digit, and the per-batch analysis of the reference's loop.
//
// The reference sorts all cells by V (np.argsort, :512) and walks them in batches of
// config.gp_batch_size cells: a batch whose unsafe tail can be accepted through local refinement
// (n_req <= max_refinement for every cell behind the first unsafe one, and the run's largest decrease
// below every cell's refined threshold) lets the loop go on, the first batch that cannot ends it.
// Every batch can be judged on its own from per-cell data in sorted order, so here
//   1. sl_adaptive_pack writes one row per cell: [vbits(V), index, decrease, -|L_v|(1+L_f), prior
//      refinement, flags] (the decrease / threshold come from one sweep with tau = 1),
//   2. rows travel to the rank that owns their sorted positions (splitters from the device radix
//      select, sl_adaptive_dest + sl_partition_by_digit + sl_gather_rows, one all-to-all),
//   3. sl_sort_pairs orders them (eight stable 8-bit passes on the value bits; rows arrive in
//      ascending index order, so ties keep the oracle's (V, index) order),
//   4. sl_adaptive_analyse judges every batch in parallel (one workgroup per batch) and reduces the
//      first batch that ends the loop, sl_adaptive_apply writes safe / refinement per cell,
//   5. rows return to the owners of the cells, sl_adaptive_scatter fills the shard's bit mask and
//      refinement array.
// Nothing of grid size is replicated across ranks.
#include "sl_common.h"

namespace {

constexpr int RS_THREADS = 256, RS_ROUNDS = 8, RS_TILE = RS_THREADS * RS_ROUNDS, RS_WAVES = RS_THREADS / 64;
constexpr int RS_MAX_BLOCKS = 2048;
constexpr int ROW = 6;                       // words of a packed cell row
enum { W_VBITS = 0, W_INDEX = 1, W_DEC = 2, W_THR0 = 3, W_REF = 4, W_FLAGS = 5 };
enum { F_INIT = 1, F_PRIOR_SAFE = 2 };

struct Segments { int nblocks; int64_t seg; };
Segments segments(int64_t n) {
    Segments s;
    int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
    int64_t per = (tiles + RS_MAX_BLOCKS - 1) / RS_MAX_BLOCKS;
    if (per < 1) per = 1;
    s.seg = per * RS_TILE;
    s.nblocks = (int)((n + s.seg - 1) / s.seg);
    if (s.nblocks < 1) s.nblocks = 1;
    return s;
}

// digit of element i: byte `pass` of keys[i] (MODE 0) or digits[i] (MODE 1)
template <int MODE>
__device__ __forceinline__ unsigned digit_of(const uint64_t* __restrict__ keys,
                                             const uint8_t* __restrict__ digits, int pass, int64_t i) {
    if (MODE == 0) return (unsigned)((keys[i] >> (8 * pass)) & 0xff);
    return digits[i];
}

// counts[digit * nblocks + block] = keys of this block's segment with that digit
template <int MODE>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(
    int64_t n, int64_t seg, const uint64_t* __restrict__ keys, const uint8_t* __restrict__ digits,
    int pass, uint32_t* __restrict__ counts) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t begin = (int64_t)blockIdx.x * seg;
    int64_t end = begin + seg;
    if (end > n) end = n;
    for (int64_t i = begin + threadIdx.x; i < end; i += RS_THREADS)
        atomicAdd(&lh[digit_of<MODE>(keys, digits, pass, i)], 1u);
    __syncthreads();
    counts[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = lh[threadIdx.x];
}

// exclusive scan of the digit-major table in place; totals[d] = keys with digit d
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t* __restrict__ counts, int nblocks,
                                                    int64_t* __restrict__ totals) {
    __shared__ uint32_t row_sum[256];
    __shared__ uint32_t row_base[256];
    uint32_t* row = counts + (int64_t)threadIdx.x * nblocks;
    uint32_t s = 0;
    for (int b = 0; b < nblocks; ++b) s += row[b];
    row_sum[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int d = 0; d < 256; ++d) { row_base[d] = acc; acc += row_sum[d]; }
    }
    __syncthreads();
    uint32_t acc = row_base[threadIdx.x];
    for (int b = 0; b < nblocks; ++b) { const uint32_t c = row[b]; row[b] = acc; acc += c; }
    if (totals) totals[threadIdx.x] = (int64_t)s;
}

// stable scatter of one pass: the element at tile position (wave, round, lane) goes to
// base[digit] + (same digit in earlier waves) + (same digit earlier in this wave)
template <int MODE, bool IOTA>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(
    int64_t n, int64_t seg, const uint64_t* __restrict__ keys, const int64_t* __restrict__ vals,
    const uint8_t* __restrict__ digits, int pass, const uint32_t* __restrict__ counts,
    uint64_t* __restrict__ keys_out, int64_t* __restrict__ vals_out) {
    __shared__ uint32_t base[256];
    __shared__ uint32_t wcount[RS_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    base[tid] = counts[(int64_t)tid * gridDim.x + blockIdx.x];
    const int64_t begin = (int64_t)blockIdx.x * seg;
    int64_t end = begin + seg;
    if (end > n) end = n;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t tile = begin; tile < end; tile += RS_TILE) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) wcount[w][tid] = 0;
        __syncthreads();
        uint64_t k[RS_ROUNDS];
        int64_t v[RS_ROUNDS];
        unsigned d[RS_ROUNDS], rank[RS_ROUNDS];
        bool valid[RS_ROUNDS];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const int64_t i = tile + (int64_t)(wave * RS_ROUNDS + r) * 64 + lane;
            valid[r] = i < end;
            k[r] = (valid[r] && (MODE == 0 || keys_out)) ? keys[i] : 0ull;
            v[r] = valid[r] ? (IOTA ? i : vals[i]) : 0;
            d[r] = valid[r] ? digit_of<MODE>(keys, digits, pass, i) : 0u;
        }
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            // lanes of this round with the same digit
            uint64_t same = __ballot(valid[r]);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d[r] >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                same &= bit ? bal : ~bal;
            }
            const unsigned before = valid[r] ? wcount[wave][d[r]] : 0u;     // earlier rounds of this wave
            rank[r] = before + (unsigned)__popcll(same & lt_mask);
            __builtin_amdgcn_wave_barrier();
            if (valid[r] && (same & lt_mask) == 0ull)                   // the group's first lane
                wcount[wave][d[r]] = before + (unsigned)__popcll(same);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // thread = digit: offsets of the waves, then the running base moves past the tile
        {
            uint32_t acc = base[tid];
            uint32_t off[RS_WAVES];
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) { off[w] = acc; acc += wcount[w][tid]; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) wcount[w][tid] = off[w];
            base[tid] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            if (valid[r]) {
                const int64_t pos = (int64_t)wcount[wave][d[r]] + rank[r];
                if (keys_out) keys_out[pos] = k[r];
                vals_out[pos] = v[r];
            }
        }
        __syncthreads();
    }
}

template <int MODE, bool IOTA>
int radix_pass(sl_ctx* ctx, int64_t n, const uint64_t* keys, const int64_t* vals, const uint8_t* digits,
               int pass, uint32_t* counts, uint64_t* keys_out, int64_t* vals_out, int64_t* totals) {
    const Segments s = segments(n);
    hipLaunchKernelGGL((k_radix_hist<MODE>), dim3(s.nblocks), dim3(RS_THREADS), 0, ctx->stream, n, s.seg,
                       keys, digits, pass, counts);
    hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(256), 0, ctx->stream, counts, s.nblocks, totals);
    hipLaunchKernelGGL((k_radix_scatter<MODE, IOTA>), dim3(s.nblocks), dim3(RS_THREADS), 0, ctx->stream, n,
                       s.seg, keys, vals, digits, pass, counts, keys_out, vals_out);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

// ---- rows ----------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t bits_of(double v) { union { double d; int64_t i; } c; c.d = v; return c.i; }
__device__ __forceinline__ double double_of(int64_t i) { union { double d; int64_t i; } c; c.i = i; return c.d; }

__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_pack(
    int64_t lo, int64_t hi, const double* __restrict__ values, const double* __restrict__ records,
    int stride, const uint64_t* __restrict__ init_bits, const uint64_t* __restrict__ prior_bits,
    const int64_t* __restrict__ prior_ref, int64_t* __restrict__ rows) {
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t i = idx - lo;
        const bool init = init_bits && ((init_bits[i >> 6] >> (i & 63)) & 1ull);
        // can_shrink: the state starts from the initial set (lyapunov.py:498-505), else from the
        // previous safe set and refinement (:506-510)
        const bool prior = prior_bits ? ((prior_bits[i >> 6] >> (i & 63)) & 1ull) : init;
        int64_t* row = rows + i * ROW;
        row[W_VBITS] = (int64_t)sl_vbits(values[i]);
        row[W_INDEX] = idx;
        row[W_DEC] = bits_of(records[i * stride]);
        row[W_THR0] = bits_of(records[i * stride + 1]);
        row[W_REF] = prior_bits ? (prior_ref ? prior_ref[i] : (prior ? 1 : 0)) : (init ? 1 : 0);
        row[W_FLAGS] = (init ? F_INIT : 0) | (prior ? F_PRIOR_SAFE : 0);
    }
}

// dest = number of splitters that are <= the row's (vbits, index) key
__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_dest(
    int64_t count, const int64_t* __restrict__ rows, const sl_select_state* __restrict__ splitters,
    int nsplit, uint8_t* __restrict__ dest) {
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        const uint64_t vb = (uint64_t)rows[i * ROW + W_VBITS];
        const int64_t idx = rows[i * ROW + W_INDEX];
        int dst = 0;
        for (int s = 0; s < nsplit; ++s) {
            const sl_key k = splitters[s].key;          // KEY_NONE when the position lies beyond the grid
            if (!sl_key_less(vb, idx, k.vbits, k.index)) dst = s + 1;
        }
        dest[i] = (uint8_t)dst;
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_gather_rows(int64_t count, int words,
                                                          const int64_t* __restrict__ perm,
                                                          const int64_t* __restrict__ in,
                                                          int64_t* __restrict__ out) {
    const int64_t total = count * words;
    for (int64_t e = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t r = e / words;
        const int w = (int)(e - r * words);
        out[e] = in[perm[r] * words + w];
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_sort_keys(int64_t m, const int64_t* __restrict__ rows,
                                                                 uint64_t* __restrict__ keys,
                                                                 int64_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * SL_BLOCK) {
        keys[i] = (uint64_t)rows[i * ROW + W_VBITS];
        vals[i] = i;
    }
}

// ---- one batch of the reference's loop -------------------------------------------------------------
struct CellView {
    double dec, thr0, thr;
    bool neg, init, prior;
    int64_t prior_ref;
    double nreq;                 // refinement the cell asks for (1 for cells that need none)
};

__device__ __forceinline__ CellView view_cell(const int64_t* __restrict__ rows, int64_t slot, double tau,
                                              double safety_factor) {
    CellView c;
    const int64_t* row = rows + slot * ROW;
    c.dec = double_of(row[W_DEC]);
    c.thr0 = double_of(row[W_THR0]);
    c.thr = c.thr0 * tau;                                    // threshold(x, tau), lyapunov.py:282-288
    c.neg = c.dec < c.thr;                                   // :441
    c.init = row[W_FLAGS] & F_INIT;
    c.prior = row[W_FLAGS] & F_PRIOR_SAFE;
    c.prior_ref = row[W_REF];
    // n with dv < threshold(tau / n): ceil(max(nan -> 0 (safety_factor threshold / decrease), 0))  :447-455
    const double scaled = safety_factor * c.thr;
    double ratio = scaled / c.dec;
    if (ratio != ratio) ratio = 0.0;
    ratio = ratio > 0.0 ? ratio : 0.0;
    c.nreq = (c.neg || c.init) ? 1.0 : ceil(ratio);          // :553-557
    return c;
}

// block-wide min of a position (threads pass INT64_MAX when they have none)
__device__ __forceinline__ int64_t block_min(int64_t v, int64_t* scratch) {
    for (int off = 32; off >= 1; off >>= 1) {
        const int64_t o = __shfl_xor((long long)v, off, 64);
        v = o < v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    int64_t r = scratch[0];
    for (int w = 1; w < SL_BLOCK / 64; ++w) r = scratch[w] < r ? scratch[w] : r;
    return r;
}
__device__ __forceinline__ double block_max(double v, double* scratch) {
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = scratch[0];
    for (int w = 1; w < SL_BLOCK / 64; ++w) r = scratch[w] > r ? scratch[w] : r;
    return r;
}

struct BatchInfo { int32_t passes, bound, stop, refine_bound; };

// judgement of batch (blockIdx.x + batch0): where the first unsafe cell sits, how far the unsafe
// tail can be refined, whether the loop goes on behind this batch
__device__ __forceinline__ BatchInfo judge_batch(int64_t q_lo, int64_t q_hi, const int64_t* __restrict__ rows,
                                                 const int64_t* __restrict__ order, double tau,
                                                 double safety_factor, double max_refinement,
                                                 int64_t* si, double* sd) {
    const int64_t NONE = INT64_MAX;
    // first cell that is neither safe before nor passes the check                     :531-537
    int64_t first = NONE;
    for (int64_t q = q_lo + threadIdx.x; q < q_hi; q += SL_BLOCK) {
        const CellView c = view_cell(rows, order[q], tau, safety_factor);
        if (!(c.prior || c.neg)) { first = q; break; }
    }
    const int64_t bound_q = block_min(first, si);
    BatchInfo info;
    if (bound_q == NONE) { info.passes = 1; info.bound = 0; info.stop = 0; info.refine_bound = 0; return info; }
    // first cell at or behind it whose required refinement is not in [1, max_refinement]   :559-568
    int64_t bad = NONE;
    for (int64_t q = bound_q + threadIdx.x; q < q_hi; q += SL_BLOCK) {
        const CellView c = view_cell(rows, order[q], tau, safety_factor);
        if (!(c.nreq >= 1.0 && c.nreq <= max_refinement)) { bad = q; break; }
    }
    int64_t stop_q = block_min(bad, si);
    if (stop_q == NONE) stop_q = q_hi;
    int64_t refine_q = bound_q;
    if (stop_q > bound_q) {
        // the refined check compares EVERY decrease of the run with the cell's refined threshold
        // (:470-474: `decrease` is the tensor of the whole fed run), i.e. the run's largest one
        double worst = -INFINITY;
        for (int64_t q = bound_q + threadIdx.x; q < stop_q; q += SL_BLOCK) {
            const CellView c = view_cell(rows, order[q], tau, safety_factor);
            const double dec = (c.dec != c.dec) ? INFINITY : c.dec;
            worst = dec > worst ? dec : worst;
        }
        worst = block_max(worst, sd);
        int64_t fail = NONE;
        for (int64_t q = bound_q + threadIdx.x; q < stop_q; q += SL_BLOCK) {
            const CellView c = view_cell(rows, order[q], tau, safety_factor);
            const double refined = c.thr0 * (tau / c.nreq);                          // :469
            if (!(worst < refined)) { fail = q; break; }
        }
        refine_q = block_min(fail, si);
        if (refine_q == NONE) refine_q = stop_q;
    }
    info.bound = (int32_t)(bound_q - q_lo);
    info.stop = (int32_t)(stop_q - bound_q);
    info.refine_bound = (int32_t)(refine_q - bound_q);
    info.passes = (stop_q == q_hi && refine_q == stop_q) ? 1 : 0;                      // :580
    return info;
}

__global__ void k_set_i64(int64_t* p, int64_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_analyse(
    int64_t m, int64_t pos0, int64_t batch, const int64_t* __restrict__ rows,
    const int64_t* __restrict__ order, double tau, double safety_factor, double max_refinement,
    int32_t* __restrict__ info_out, int64_t* __restrict__ first_break) {
    __shared__ int64_t si[SL_BLOCK / 64];
    __shared__ double sd[SL_BLOCK / 64];
    const int64_t q_lo = (int64_t)blockIdx.x * batch;
    int64_t q_hi = q_lo + batch;
    if (q_hi > m) q_hi = m;
    const BatchInfo info = judge_batch(q_lo, q_hi, rows, order, tau, safety_factor, max_refinement, si, sd);
    if (threadIdx.x == 0) {
        int32_t* o = info_out + 4 * (int64_t)blockIdx.x;
        o[0] = info.passes; o[1] = info.bound; o[2] = info.stop; o[3] = info.refine_bound;
        if (!info.passes) atomicMin((long long*)first_break, (long long)(pos0 / batch + blockIdx.x));
    }
}

// what the loop leaves in safe_batch / refine_batch of this batch: batches before the one that
// ends the loop are fully accepted, that one is cut behind bound + refine_bound (:581-582), later
// ones are never touched
__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_apply(
    int64_t m, int64_t pos0, int64_t batch, const int64_t* __restrict__ rows,
    const int64_t* __restrict__ order, const int32_t* __restrict__ info_in, double tau,
    double safety_factor, int64_t b_star, int64_t* __restrict__ out_rows) {
    const int64_t q_lo = (int64_t)blockIdx.x * batch;
    int64_t q_hi = q_lo + batch;
    if (q_hi > m) q_hi = m;
    const int64_t b = pos0 / batch + blockIdx.x;
    const int32_t* in = info_in + 4 * (int64_t)blockIdx.x;
    const bool all_safe = in[0] && in[2] == 0 && in[3] == 0 && in[1] == 0;   // no unsafe cell at all
    const int64_t bound_q = q_lo + in[1], cut_q = bound_q + in[3];
    for (int64_t q = q_lo + threadIdx.x; q < q_hi; q += SL_BLOCK) {
        const int64_t slot = order[q];
        const CellView c = view_cell(rows, slot, tau, safety_factor);
        bool safe;
        int64_t ref;
        if (b > b_star) {
            safe = c.prior; ref = c.prior_ref;
        } else {
            // step 1 of every processed batch: safe |= negative, refinement[negative] = 1   :533-535
            safe = c.prior || c.neg;
            ref = c.neg ? 1 : c.prior_ref;
            // a batch with an unsafe cell also resets the known-safe cells of the WHOLE batch:
            // refine_batch[negative | initial] = 1                                        :553-557
            if (!all_safe && c.init) ref = 1;
            // "all_safe" has to be told from "first unsafe cell at position 0": passes && stop == 0
            // && refine_bound == 0 && bound == 0 also describes an EMPTY refinement of a batch whose
            // first cell is unsafe, but such a batch does not pass.
            if (!all_safe && q >= bound_q) {
                if (q < cut_q) { safe = true; ref = (int64_t)c.nreq; }
                else if (b == b_star) { safe = false; ref = 0; }
                else { safe = true; ref = (int64_t)c.nreq; }     // unreachable: a passing batch has cut_q == q_hi
            }
        }
        int64_t* o = out_rows + slot * 3;
        o[0] = rows[slot * ROW + W_INDEX];
        o[1] = safe ? 1 : 0;
        o[2] = ref;
    }
}

// rows [index, safe, refinement] of this rank's cells -> the shard's bit mask and refinement array
__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_scatter(
    int64_t lo, int64_t m, const int64_t* __restrict__ out_rows, uint64_t* __restrict__ safe_bits,
    int64_t* __restrict__ refinement) {
    for (int64_t j = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; j < m;
         j += (int64_t)gridDim.x * SL_BLOCK) {
        const int64_t i = out_rows[3 * j] - lo;
        if (out_rows[3 * j + 1]) atomicOr((unsigned long long*)&safe_bits[i >> 6], 1ull << (i & 63));
        refinement[i] = out_rows[3 * j + 2];
    }
}

// the initial safe set is kept: safe, refinement 1                                       :601-606
__global__ __launch_bounds__(SL_BLOCK) void k_adaptive_keep_init(
    int64_t count, const uint64_t* __restrict__ init_bits, uint64_t* __restrict__ safe_bits,
    int64_t* __restrict__ refinement, int64_t* __restrict__ n_safe) {
    int64_t local = 0;
    const int64_t nwords = (count + 63) >> 6;
    for (int64_t w = (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; w < nwords;
         w += (int64_t)gridDim.x * SL_BLOCK) {
        uint64_t init = init_bits ? init_bits[w] : 0ull;
        const uint64_t word = safe_bits[w] | init;
        safe_bits[w] = word;
        local += __popcll(word);
        while (init) {
            const int bit = __ffsll((long long)init) - 1;
            refinement[w * 64 + bit] = 1;
            init &= init - 1;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor((long long)local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd((unsigned long long*)n_safe, (unsigned long long)local);
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int sl_sort_pairs(sl_ctx* ctx, int64_t n, uint64_t* d_keys, int64_t* d_vals,
                             uint64_t* d_keys_tmp, int64_t* d_vals_tmp, uint32_t* d_counts) {
    if (!ctx || n < 0 || n > 0x7fffffffll || (n && (!d_keys || !d_vals || !d_keys_tmp || !d_vals_tmp || !d_counts)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_sort_pairs: bad argument");
    if (n == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t* ka = d_keys; int64_t* va = d_vals;
    uint64_t* kb = d_keys_tmp; int64_t* vb = d_vals_tmp;
    for (int pass = 0; pass < 8; ++pass) {             // an even number of passes: the result ends in d_keys / d_vals
        int rc = radix_pass<0, false>(ctx, n, ka, va, nullptr, pass, d_counts, kb, vb, nullptr);
        if (rc) return rc;
        uint64_t* tk = ka; ka = kb; kb = tk;
        int64_t* tv = va; va = vb; vb = tv;
    }
    return SL_OK;
}

extern "C" int sl_partition_by_digit(sl_ctx* ctx, int64_t n, const uint8_t* d_digits, int64_t* d_perm,
                                     int64_t* d_bucket_counts, uint32_t* d_counts) {
    if (!ctx || n < 0 || n > 0x7fffffffll || !d_bucket_counts || (n && (!d_digits || !d_perm || !d_counts)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_partition_by_digit: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n == 0) {
        SL_HIP_CHECK(ctx, hipMemsetAsync(d_bucket_counts, 0, 256 * sizeof(int64_t), ctx->stream));
        return SL_OK;
    }
    return radix_pass<1, true>(ctx, n, nullptr, nullptr, d_digits, 0, d_counts, nullptr, d_perm, d_bucket_counts);
}

extern "C" int sl_gather_rows(sl_ctx* ctx, int64_t count, int words, const int64_t* d_perm,
                              const int64_t* d_rows_in, int64_t* d_rows_out) {
    if (!ctx || count < 0 || words < 1 || (count && (!d_perm || !d_rows_in || !d_rows_out)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_gather_rows: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_gather_rows, dim3(sl_grid_blocks(count * words)), dim3(SL_BLOCK), 0, ctx->stream,
                       count, words, d_perm, d_rows_in, d_rows_out);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_pack(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                                const double* d_records, int record_stride,
                                const uint64_t* d_init_bits, const uint64_t* d_prior_bits,
                                const int64_t* d_prior_ref, int64_t* d_rows) {
    if (!ctx || lo < 0 || hi < lo || record_stride < 2 || (hi > lo && (!d_values || !d_records || !d_rows)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_pack: bad argument");
    if (hi == lo) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_adaptive_pack, dim3(sl_grid_blocks(hi - lo)), dim3(SL_BLOCK), 0, ctx->stream, lo,
                       hi, d_values, d_records, record_stride, d_init_bits, d_prior_bits, d_prior_ref, d_rows);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_dest(sl_ctx* ctx, int64_t count, const int64_t* d_rows,
                                const sl_select_state* d_splitters, int nsplit, uint8_t* d_dest) {
    if (!ctx || count < 0 || nsplit < 0 || nsplit > 255 || (count && (!d_rows || !d_dest)) ||
        (nsplit && !d_splitters))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_dest: bad argument");
    if (count == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_adaptive_dest, dim3(sl_grid_blocks(count)), dim3(SL_BLOCK), 0, ctx->stream, count,
                       d_rows, d_splitters, nsplit, d_dest);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_sort_keys(sl_ctx* ctx, int64_t m, const int64_t* d_rows, uint64_t* d_keys,
                                     int64_t* d_vals) {
    if (!ctx || m < 0 || (m && (!d_rows || !d_keys || !d_vals)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_sort_keys: bad argument");
    if (m == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_adaptive_sort_keys, dim3(sl_grid_blocks(m)), dim3(SL_BLOCK), 0, ctx->stream, m,
                       d_rows, d_keys, d_vals);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_analyse(sl_ctx* ctx, int64_t m, int64_t pos0, int64_t batch,
                                   const int64_t* d_rows, const int64_t* d_order, double tau,
                                   double safety_factor, int64_t max_refinement, int32_t* d_info,
                                   int64_t* d_first_break) {
    if (!ctx || m < 0 || batch < 1 || pos0 < 0 || (pos0 % batch) || !d_first_break ||
        (m && (!d_rows || !d_order || !d_info)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_analyse: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(64), 0, ctx->stream, d_first_break, INT64_MAX);
    if (m) {
        const int64_t nb = (m + batch - 1) / batch;
        hipLaunchKernelGGL(k_adaptive_analyse, dim3((unsigned)nb), dim3(SL_BLOCK), 0, ctx->stream, m, pos0,
                           batch, d_rows, d_order, tau, safety_factor, (double)max_refinement, d_info,
                           d_first_break);
    }
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_apply(sl_ctx* ctx, int64_t m, int64_t pos0, int64_t batch,
                                 const int64_t* d_rows, const int64_t* d_order, const int32_t* d_info,
                                 double tau, double safety_factor, int64_t b_star, int64_t* d_out_rows) {
    if (!ctx || m < 0 || batch < 1 || pos0 < 0 || (pos0 % batch) ||
        (m && (!d_rows || !d_order || !d_info || !d_out_rows)))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_apply: bad argument");
    if (m == 0) return SL_OK;
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t nb = (m + batch - 1) / batch;
    hipLaunchKernelGGL(k_adaptive_apply, dim3((unsigned)nb), dim3(SL_BLOCK), 0, ctx->stream, m, pos0, batch,
                       d_rows, d_order, d_info, tau, safety_factor, b_star, d_out_rows);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}

extern "C" int sl_adaptive_scatter(sl_ctx* ctx, int64_t lo, int64_t hi, int64_t m, const int64_t* d_out_rows,
                                   const uint64_t* d_init_bits, uint64_t* d_safe_bits,
                                   int64_t* d_refinement, int64_t* d_safe_count) {
    if (!ctx || lo < 0 || hi < lo || m < 0 || !d_safe_count ||
        (hi > lo && (!d_safe_bits || !d_refinement)) || (m && !d_out_rows))
        return sl_fail(ctx, SL_ERR_INVALID, "sl_adaptive_scatter: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipMemsetAsync(d_safe_count, 0, sizeof(int64_t), ctx->stream));
    if (hi == lo) return SL_OK;
    SL_HIP_CHECK(ctx, hipMemsetAsync(d_safe_bits, 0, sizeof(uint64_t) * (size_t)((hi - lo + 63) / 64), ctx->stream));
    if (m)
        hipLaunchKernelGGL(k_adaptive_scatter, dim3(sl_grid_blocks(m)), dim3(SL_BLOCK), 0, ctx->stream, lo, m,
                           d_out_rows, d_safe_bits, d_refinement);
    hipLaunchKernelGGL(k_adaptive_keep_init, dim3(sl_grid_blocks((hi - lo + 63) / 64)), dim3(SL_BLOCK), 0,
                       ctx->stream, hi - lo, d_init_bits, d_safe_bits, d_refinement, d_safe_count);
    SL_HIP_CHECK(ctx, hipGetLastError());
    return SL_OK;
}
