/*
 * sl_hip.h - C ABI of libslhip.so, the MI355X (gfx950) engine under safe_learning_amd.
 *
 * The reference (befelix/safe_learning) has no FFI: its hot path is a Python loop that
 * feeds 10 000-cell batches to a TensorFlow graph (safe_learning/lyapunov.py:517-587,
 * safe_learning/reinforcement_learning.py:65-140, 213-279).  This header is the boundary a
 * maintainer would bind (ctypes stub in INTEGRATION.md) to replace exactly those loops; each
 * entry point cites the reference code it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every function returns 0 (SL_OK) or a
 *    negative error code, never throws, never aborts; sl_last_error() gives the message.
 *  - "h_" pointers are host memory, copied before the call returns.  "d_" pointers are
 *    device memory owned by the caller (e.g. a torch tensor's data_ptr()) on the context's
 *    device and must stay valid until the context's stream has executed the call.
 *  - All launches go to the stream given at sl_ctx_create (a hipStream_t; NULL = the
 *    default stream) and are asynchronous unless stated otherwise.
 *  - All arithmetic is float64 (safe_learning/configuration.py:16); masks are bit masks,
 *    bit (i % 64) of word (i / 64), cell indices are flat C-order GridWorld indices
 *    (safe_learning/functions.py:622-638, 714-731).
 *  - A context is not re-entrant; use one context per GPU / per host thread.
 */
#ifndef SL_HIP_H
#define SL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libslhip.so is built with -fvisibility=hidden: only the entry points declared here are exported. */
#define SL_API __attribute__((visibility("default")))

#define SL_OK               0
#define SL_ERR_INVALID     (-1)   /* bad argument / model not set            */
#define SL_ERR_HIP         (-2)   /* a HIP runtime call failed               */
#define SL_ERR_UNSUPPORTED (-3)   /* combination not implemented             */
#define SL_ERR_NOMEM       (-4)

#define SL_MAX_STATE_DIM   6
#define SL_MAX_ACTION_DIM  2
#define SL_MAX_INPUT_DIM   8      /* state + action                          */
#define SL_MAX_GP_HEADS    6
#define SL_MAX_NN_LAYERS   4
#define SL_MAX_SIMPLICES   32     /* unit-cell simplices (Qhull gives 22 in 4-D) */

typedef struct sl_ctx sl_ctx;

/* ---- function kinds -------------------------------------------------------------- */
enum sl_policy_kind {
    SL_POLICY_LINEAR = 1,   /* u = x K^T            (LinearSystem, functions.py:1567-1583)  */
    SL_POLICY_CONST  = 2,   /* u = constant row     (reinforcement_learning.py:233-235,268) */
    SL_POLICY_TABLE  = 3,   /* u = table[i]         (per-vertex policy on the same grid)    */
    SL_POLICY_TRI    = 4,   /* u = Triangulation(x) on the auxiliary grid #1                */
    SL_POLICY_NETWORK = 5   /* u = NeuralNetwork(x)  (functions.py:1663-1729), sl_policy_network_set */
};
enum sl_dynamics_kind {
    SL_DYN_LINEAR   = 1,    /* [x,u] M^T            (functions.py:1567-1583)                */
    SL_DYN_PENDULUM = 2,    /* examples/utilities.py:242-289, 10 Euler sub-steps            */
    SL_DYN_CARTPOLE = 3,    /* examples/utilities.py:387-437, 10 Euler sub-steps            */
    SL_DYN_GP       = 4     /* GP posterior (functions.py:417-458, 507-515, 278-291)        */
};
enum sl_value_kind {
    SL_V_QUADRATIC  = 1,    /* x P x^T              (functions.py:1534-1539)                */
    SL_V_TRI        = 2,    /* Triangulation on the auxiliary grid #0 (functions.py:1473-1499) */
    SL_V_NETWORK    = 3     /* LyapunovNetwork      (examples/utilities.py:85-104)          */
};
enum sl_lipschitz_kind {
    SL_LIP_CONST       = 0, /* scalar                                  (lyapunov.py:241-263) */
    SL_LIP_ABS_LINEAR  = 1, /* |x G^T| per column  (adaptive_safety_verification.ipynb c.17) */
    SL_LIP_NORM_LINEAR = 2, /* ||x G^T||_1, one column                                       */
    SL_LIP_ABS_GRAD    = 3, /* |grad V(x)| per column (inverted_pendulum.ipynb cell 14)      */
    SL_LIP_NORM_GRAD   = 4  /* ||grad V(x)||_1, one column (lyapunov_function_learning c.19) */
};

/* ---- plain-old-data model description (copied by sl_model_set) ---------------------- */
typedef struct sl_grid_desc {            /* GridWorld: functions.py:591-620 */
    int32_t d;                           /* state dimension                 */
    int32_t reserved;
    int64_t num_points[SL_MAX_STATE_DIM];
    double  offset[SL_MAX_STATE_DIM];    /* limits[:,0]                     */
    double  unit_maxes[SL_MAX_STATE_DIM];
    double  upper[SL_MAX_STATE_DIM];     /* limits[:,1]                     */
} sl_grid_desc;

typedef struct sl_policy_desc {
    int32_t kind, m, saturate, reserved;             /* Saturation: functions.py:349-354 */
    double  matrix[SL_MAX_ACTION_DIM][SL_MAX_STATE_DIM];   /* rows = outputs            */
    double  lower[SL_MAX_ACTION_DIM], upper[SL_MAX_ACTION_DIM];
    double  constant[SL_MAX_ACTION_DIM];
    const double* d_table;                           /* [nindex][m] when kind == TABLE   */
} sl_policy_desc;

typedef struct sl_dynamics_desc {
    int32_t kind, normalize;
    double  matrix[SL_MAX_STATE_DIM][SL_MAX_INPUT_DIM];    /* LINEAR: rows = outputs; GP: prior mean */
    double  tx[SL_MAX_STATE_DIM], tx_inv[SL_MAX_STATE_DIM];
    double  tu[SL_MAX_ACTION_DIM];
    double  coef[16];                    /* analytic coefficients, see sl_model.h */
} sl_dynamics_desc;

typedef struct sl_value_desc {
    int32_t kind, negate;                /* negate: V = -f (functions.py:120-122)           */
    double  matrix[SL_MAX_INPUT_DIM][SL_MAX_INPUT_DIM];    /* QUADRATIC                     */
} sl_value_desc;

enum { SL_LF_CONST = 0, SL_LF_AFFINE_NORM1 = 1 };

typedef struct sl_lipschitz_desc {
    int32_t lv_kind, lv_cols;            /* columns of L_v(x): 1 or d                       */
    double  lv_const;
    double  lv_matrix[SL_MAX_STATE_DIM][SL_MAX_STATE_DIM]; /* rows = outputs                */
    double  lf_const;                    /* L_f (closed loop): the scalar, or the constant c */
    double  tau;
    int32_t lf_kind, lf_reserved;        /* SL_LF_CONST, or SL_LF_AFFINE_NORM1:              */
    double  lf_matrix[SL_MAX_STATE_DIM][SL_MAX_STATE_DIM]; /* L_f(x) = c + ||lf_matrix x||_1  */
} sl_lipschitz_desc;                     /* (state-dependent L_f, lyapunov.py:227-244)       */

typedef struct sl_model_desc {
    sl_grid_desc      grid;
    sl_policy_desc    policy;
    sl_dynamics_desc  dynamics;
    sl_value_desc     value;             /* Lyapunov function V / value function            */
    sl_lipschitz_desc lipschitz;
    sl_value_desc     reward;            /* QUADRATIC on [x,u] (PolicyIteration only)       */
    double            gamma;             /* discount (reinforcement_learning.py:47)          */
} sl_model_desc;

/* (V, flat index) ordering key: ascending V, ties by ascending index (oracle/np_lyapunov.py). */
typedef struct sl_key { uint64_t vbits; int64_t index; } sl_key;

/* Result block of the Lyapunov passes (device memory, 8 x 8 bytes). */
typedef struct sl_sweep_result {
    sl_key   fail;          /* lexmin{(V_i,i): not (negative_i or init_i)}; vbits=~0 if none */
    sl_key   last_safe;     /* lexmax{(V_i,i) < key_star}; index=-1 if none                  */
    sl_key   max_key;       /* lexmax over all cells of the range                            */
    int64_t  count_below;   /* #cells with key < key_star                                    */
    int64_t  count_safe;    /* #bits set in the written safe mask                            */
} sl_sweep_result;

/* ---- context ------------------------------------------------------------------------- */
SL_API int  sl_version(void);
SL_API int  sl_ctx_create(int device, void* hip_stream, sl_ctx** out);
SL_API int  sl_ctx_destroy(sl_ctx* ctx);
SL_API const char* sl_last_error(const sl_ctx* ctx);      /* ctx may be NULL: last global error */
SL_API int  sl_ctx_synchronize(sl_ctx* ctx);
/* Name (with its template arguments) of the kernel(s) the last sl_lyap_sweep / sl_bellman_sweep of
 * this context launched for the per-cell work, e.g. "k_gp_sweep4<d=4, m=1, xs_global=0>" or
 * "k_gp_sweep<...> + k_nn_check_mfma<...>" for a two-pass sweep; "" before the first sweep.  The
 * string lives in the context and is overwritten by the next sweep. */
SL_API const char* sl_last_kernel(const sl_ctx* ctx);
/* Kernel durations for benchmarks.  With slots > 0 every sl_lyap_sweep (channel 0),
 * sl_lyap_finalize_dev (channel 1) and sl_bellman_sweep (channel 2) brackets its launches with a pair
 * of HIP events on the context's stream, up to `slots` calls per channel (later calls are not
 * recorded); the events are created here, not per call.  slots = 0 switches it off and frees them.
 * sl_timing_collect waits for the recorded calls of one channel, writes their durations in
 * milliseconds to h_ms (at most `capacity`), the number written to *count, and empties the channel.
 * (The reference has no counterpart: its timings are the notebooks' wall clocks.) */
SL_API int  sl_timing_configure(sl_ctx* ctx, int slots);
SL_API int  sl_timing_collect(sl_ctx* ctx, int channel, double* h_ms, int capacity, int* count);

/* ---- model upload (copies; replaces the TF graph build of lyapunov.py:431-443) --------- */
SL_API int  sl_model_set(sl_ctx* ctx, const sl_model_desc* h_model);

/* GP head `head` of the dynamics (FunctionStack: one head per output column,
 * functions.py:278-291; a shared-kernel multi-output GPRCached is one head with dout = D).
 *   h_X      [n][p]   training inputs                     (GPRCached.X,      functions.py:382)
 *   h_Linv   [n][n]   inverse of the Cholesky factor of K + sigma_n^2 I, lower triangular,
 *                     so that a = Linv k_x equals tf.matrix_triangular_solve (functions.py:441)
 *   h_alpha  [n][dout] L^-1 (Y - m(X))                    (GPRCached.alpha,  functions.py:405-409)
 * RBF (gpflow 0.4.0): k = variance * exp(-0.5 sum_q ((x_q - x'_q) / lengthscales_q)^2). */
SL_API int  sl_gp_set_head(sl_ctx* ctx, int head, int n, int p, int dout, int col0,
                    const double* h_X, const double* h_Linv, const double* h_alpha,
                    double variance, const double* h_lengthscales);
/* The same head with a kernel of the family the reference's notebooks build from gpflow 0.4.0's
 * kernels.py (examples/inverted_pendulum.ipynb:152-158: Linear + Matern32 * Linear; the functions
 * GPRCached calls are kern.K and kern.Kdiag, functions.py:401, 438, 445, 450): a SUM of PRODUCTS
 * of leaf kernels.  Factor f belongs to product `product` (products are numbered 0, 1, ... in
 * the order of their first factor; the factors of a product are adjacent):
 *   SL_KERNEL_RBF       variance[0] exp(-r2 / 2),                    r2 = sum_q ((x_q - x'_q) inv_lengthscales[q])^2
 *   SL_KERNEL_MATERN32  variance[0] (1 + sqrt(3) r) exp(-sqrt(3) r), r = sqrt(r2 + 1e-12)   (Stationary.euclid_dist)
 *   SL_KERNEL_LINEAR    sum_q variance[q] x_q x'_q
 * Dimensions outside a leaf's active_dims carry inv_lengthscales[q] = 0 (variance[q] = 0 for a
 * Linear leaf).  K(x, x) of the posterior variance (functions.py:450) follows from the same
 * formulas (Kdiag).  The inputs are stored unscaled. */
#define SL_KERNEL_RBF 0
#define SL_KERNEL_MATERN32 1
#define SL_KERNEL_LINEAR 2
#define SL_KERNEL_MAX_FACTORS 8
typedef struct sl_gp_kernel_factor {
    int32_t kind, product;
    double  variance[SL_MAX_INPUT_DIM];
    double  inv_lengthscales[SL_MAX_INPUT_DIM];
} sl_gp_kernel_factor;
typedef struct sl_gp_kernel {
    int32_t nfactors, reserved;
    sl_gp_kernel_factor factor[SL_KERNEL_MAX_FACTORS];
} sl_gp_kernel;
SL_API int  sl_gp_set_head_kernel(sl_ctx* ctx, int head, int n, int p, int dout, int col0,
                    const double* h_X, const double* h_Linv, const double* h_alpha,
                    const sl_gp_kernel* h_kernel);
/* One more training point for an uploaded head (GaussianProcess.add_data_point,
 * functions.py:525-546) without re-packing: the rank-one extension of the cached factors touches
 * one new row of L^-1, so only that row is scattered into the fragment layout.
 *   h_x [p], h_linv_row [n+1] = row n of the extended L^-1 (lower triangle incl. the diagonal),
 *   h_alpha_new [dout] = the new last row of alpha = L^-1 (Y - m(X)).
 * alpha' = Linv^T alpha gets the rank-one term linv_row * alpha_new.  Returns SL_ERR_UNSUPPORTED
 * when the padded capacity of the head is exhausted (then call sl_gp_set_head again). */
SL_API int  sl_gp_append_point(sl_ctx* ctx, int head, const double* h_x, const double* h_linv_row,
                        const double* h_alpha_new);
SL_API int  sl_gp_configure(sl_ctx* ctx, int nheads, double beta);

/* Auxiliary grid #slot with a per-vertex table (Triangulation: functions.py:1002-1032,
 * 1064-1101): slot 0 = value function, slot 1 = policy.  h_simplices [nsimplex][d+1] are the
 * unit-cell vertices as {0,1}^d corner codes (bit k = dimension k), h_hyperplanes
 * [nsimplex][d][d] = inv(vertices[1:] - vertices[0]); d_table [nindex][ncols] stays caller-owned. */
SL_API int  sl_tri_set(sl_ctx* ctx, int slot, const sl_grid_desc* h_grid, int nsimplex,
                const int32_t* h_simplices, const double* h_hyperplanes,
                const double* h_discrete_points, int project, int ncols, const double* d_table);
/* New vertex values for a slot whose grid stays.  Call it again whenever the values of the POLICY
 * table (slot 1) change, also when they were overwritten in place: what the policy-evaluation
 * sweep derives from the policy alone is kept between sweeps until this, sl_tri_set(slot 1) or an
 * sl_model_set with another policy description says that the policy is a new one. */
SL_API int  sl_tri_set_table(sl_ctx* ctx, int slot, const double* d_table);

/* LyapunovNetwork (examples/utilities.py:85-104): h_kernels = per-layer kernel matrices
 * [out_i][in_i] concatenated; activation codes 0 = linear, 1 = tanh, 2 = relu. */
SL_API int  sl_network_set(sl_ctx* ctx, int nlayers, const int32_t* h_dims /* nlayers+1 */,
                    const int32_t* h_activations, const double* h_kernels);

/* NeuralNetwork as the policy (functions.py:1663-1729; the policies of examples/inverted_pendulum.ipynb
 * :215 and of the reinforcement-learning notebooks): a chain of dense layers
 *   net <- act_l(net W_l + b_l),  u = output_scale * net
 * h_dims [nlayers + 1] = input width (the state dimension) followed by the units of every layer (the
 * reference's `layers`), widths <= 64, the last <= SL_MAX_ACTION_DIM; h_activations [nlayers]: 0 none,
 * 1 tanh, 2 relu, 3 sigmoid; h_kernels = the W_l ([in][out] row-major, tf.layers.dense's kernel)
 * concatenated; h_has_bias [nlayers] (may be NULL: none) and h_biases = the b_l of the layers that
 * have one, concatenated (the reference's output layer never has).  With policy.kind =
 * SL_POLICY_NETWORK the entry points that evaluate the policy (sl_lyap_sweep, sl_bellman_sweep with
 * n_actions == 0, sl_eval_points) evaluate the network once per cell / vertex / point into an
 * action table of their own and run the kernels on that table (policy.saturate applies as usual). */
SL_API int  sl_policy_network_set(sl_ctx* ctx, int nlayers, const int32_t* h_dims,
                           const int32_t* h_activations, const double* h_kernels,
                           const double* h_biases, const int32_t* h_has_bias, double output_scale);

/* ---- Lyapunov passes ------------------------------------------------------------------- */
/* values[i-lo] = V(all_points[i]), i in [lo,hi): the points of functions.py:622-638 (np.linspace:
 * the last point of each dimension is exactly the upper limit).  Replaces
 * Lyapunov.update_values (lyapunov.py:305-322). */
SL_API int  sl_values(sl_ctx* ctx, int64_t lo, int64_t hi, double* d_values);

/* Decrease check of every cell i in [lo,hi) (lo % 64 == 0): policy -> dynamics ->
 * V(f(x)) - V(x) + L_v.err < -|L_v|_1 (1 + L_f) tau  (lyapunov.py:436-441, 265-288, 324-376).
 * Replaces the batch loop lyapunov.py:524-587.
 *   d_init_bits   in  (may be NULL) cells that count as safe without a check
 *   d_values      in  V on the grid from sl_values: the ordering keys of lyapunov.py:512
 *                     (may be NULL: the V(x_i) computed for the decrease is used instead)
 *   d_neg_bits    out the `negative` mask
 *   d_result      out ->fail = lexmin over failing cells (other fields untouched)
 *   d_dbg         out (may be NULL) per cell [decrease, threshold, mean[d], err[d]] */
SL_API int  sl_lyap_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, const uint64_t* d_init_bits,
                   const double* d_values, uint64_t* d_neg_bits, sl_sweep_result* d_result,
                   double* d_dbg);

/* safe_i = init_i | (key_i < key_star) | (prev_i & key_i >= key_keep), the parallel form of
 * lyapunov.py:513-606.  d_prev_bits may be NULL.  Fills last_safe, max_key,
 * count_below, count_safe of d_result. */
SL_API int  sl_lyap_finalize(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                      const uint64_t* d_init_bits, const uint64_t* d_prev_bits,
                      sl_key key_star, sl_key key_keep, uint64_t* d_safe_bits,
                      sl_sweep_result* d_result);

/* One radix-select pass over the (V, index) keys of [lo,hi): 256-bin histogram of byte
 * `byte` (7 = most significant) of the vbits (which = 0) or of the index (which = 1, only cells
 * with vbits == prefix) among keys whose higher bytes equal `prefix`'s.  d_hist[256] is ADDED to
 * (zero it first).  Gives the k-th order statistic that lyapunov.py:590-595 reads through argsort. */
SL_API int  sl_select_pass(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values, int which,
                    int byte, uint64_t prefix, uint64_t vbits_equal, uint64_t* d_hist);

/* ---- the same rule with every decision read from DEVICE memory (no host round trip between the
 * passes of one update_safe_set; the multi-GPU path of SURVEY.md 8e) ------------------------------ */

/* *out = 1 when d_values may be NULL in sl_lyap_sweep AND in the three passes below: V is a
 * QuadraticFunction (functions.py:1503-1539) on a grid of 1..4 dimensions (last axis a multiple of 8
 * cells) whose np.linspace points
 * (functions.py:612-638, the points Lyapunov.update_values evaluates V on, lyapunov.py:321) equal
 * index_to_state (functions.py:728-731) bit for bit.  The ordering keys of lyapunov.py:512 are then
 * recomputed from the cell index (8 cells of a grid row per thread, the prefix of the ordered sums
 * shared) instead of read: the values array (2.1 GB at 128^4) need not exist. */
SL_API int  sl_values_implicit(sl_ctx* ctx, int* out);

/* d_out = fold of `count` result records that lie in device memory (e.g. every rank's record after
 * an all-gather): lexmin of `fail`, lexmax of `last_safe` and `max_key`, sums of the counters -
 * the reductions np.argsort + the batch loop of lyapunov.py:512-595 perform on one host. */
SL_API int  sl_fold_results(sl_ctx* ctx, const sl_sweep_result* d_records, int count,
                     sl_sweep_result* d_out);

/* sl_lyap_finalize with key_star = d_folded->fail and key_keep = *d_keep (NULL: none) read by the
 * kernel.  d_values may be NULL (sl_values_implicit).  d_result: ->fail is copied from d_folded,
 * the other fields are this range's statistics (fold them across ranks with sl_fold_results). */
SL_API int  sl_lyap_finalize_dev(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                          const uint64_t* d_init_bits, const uint64_t* d_prev_bits,
                          const sl_sweep_result* d_folded, const sl_key* d_keep,
                          uint64_t* d_safe_bits, sl_sweep_result* d_result);

/* The refinement array N(x) of lyapunov.py:220-225 through a NON-adaptive update with
 * can_shrink = False (lyapunov.py:507-510, 531, 585-587, 601-606), in place on this range's cells:
 *   ref_i <- init_i ? 1 : key_i < key* ? (negative_i ? 1 : ref_i) : (key_i >= key_keep ? ref_i : 0)
 * with key* = d_folded->fail, key_keep = *d_keep (NULL: no failure, nothing is cleared), d_neg_bits the
 * sweep's `negative` words.  Only needed when an adaptive update left values other than 0 / 1. */
SL_API int  sl_refinement_carry(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                         const uint64_t* d_init_bits, const uint64_t* d_neg_bits,
                         const sl_sweep_result* d_folded, const sl_key* d_keep, int64_t* d_refinement);

/* Radix select of the k-th smallest (V, index) key over all ranks (0-based; what
 * values[order[k]] of lyapunov.py:512, 590-595 reads) with its state in device memory. */
typedef struct sl_select_state {
    uint64_t prefix;        /* digits found so far (of vbits, then of the index)                 */
    int64_t  remaining;     /* rank among the keys that share the prefix                         */
    sl_key   key;           /* the result once both phases are through; KEY_NONE if none         */
    int64_t  rank;          /* the k the select was started with                                 */
    int64_t  none;          /* 1: k >= n_total (no such key), every pass is a no-op              */
    int64_t  pad[2];
} sl_select_state;          /* 64 bytes */

/* k >= 0: select that rank.  k < 0: rank = (d_folded->count_below / batch + 1) * batch, the first
 * position behind the batch that holds the first failure (lyapunov.py:585-587: later batches keep
 * their previous state).  n_total = cells of the whole grid. */
SL_API int  sl_select_begin(sl_ctx* ctx, sl_select_state* d_state, int64_t k, int64_t batch,
                     const sl_sweep_result* d_folded, int64_t n_total);
/* Histogram of one byte (7 = most significant first) of the keys of [lo,hi) that match the state's
 * prefix: which = 0 the vbits, which = 1 the index among cells with vbits == state->key.vbits.
 * d_hist[256] is zeroed and filled; SUM it over the ranks, then call sl_select_digit. */
SL_API int  sl_select_hist(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values, int which,
                    int byte, const sl_select_state* d_state, uint64_t* d_hist);
SL_API int  sl_select_digit(sl_ctx* ctx, int which, int byte, const uint64_t* d_hist,
                     sl_select_state* d_state);

/* ---- the adaptive branch of update_safe_set (lyapunov.py:445-487, 540-582) -------------------------
 * The reference walks the cells in ascending-V order in batches of config.gp_batch_size; a batch whose
 * unsafe tail can be accepted through local refinement lets the loop go on, the first one that
 * cannot ends it.  Every batch is judged on its own from per-cell rows in sorted order. */

/* Stable ascending sort of n (key, value) pairs by key: eight 8-bit radix passes (np.argsort of
 * lyapunov.py:512 with ties in input order).  Result in d_keys / d_vals; d_keys_tmp / d_vals_tmp [n]
 * and d_counts [SL_SORT_COUNT_WORDS] are scratch. */
#define SL_SORT_COUNT_WORDS (256 * 2048)
SL_API int  sl_sort_pairs(sl_ctx* ctx, int64_t n, uint64_t* d_keys, int64_t* d_vals,
                   uint64_t* d_keys_tmp, int64_t* d_vals_tmp, uint32_t* d_counts);
/* Stable partition: d_perm = the positions 0..n-1 ordered by d_digits[position] (positions with
 * equal digits keep their order); d_bucket_counts[256] = elements per digit. */
SL_API int  sl_partition_by_digit(sl_ctx* ctx, int64_t n, const uint8_t* d_digits, int64_t* d_perm,
                           int64_t* d_bucket_counts, uint32_t* d_counts);
/* d_rows_out[r] = d_rows_in[d_perm[r]] for rows of `words` 8-byte words. */
SL_API int  sl_gather_rows(sl_ctx* ctx, int64_t count, int words, const int64_t* d_perm,
                    const int64_t* d_rows_in, int64_t* d_rows_out);

/* One row of SL_ADAPTIVE_ROW_WORDS words per cell of [lo,hi): vbits(V), flat index, decrease and
 * threshold(x, tau = 1) = -|L_v(x)|_1 (1 + L_f(x)) from the records of a sweep run with tau = 1
 * (d_records: [hi-lo][record_stride], columns 0 and 1), the refinement N(x) and the flags the loop
 * starts from: d_prior_bits = NULL (can_shrink: the initial set, lyapunov.py:498-505) or the
 * previous safe set with d_prior_ref = the previous refinement (NULL: 1 on safe cells) (:506-510). */
#define SL_ADAPTIVE_ROW_WORDS 6
SL_API int  sl_adaptive_pack(sl_ctx* ctx, int64_t lo, int64_t hi, const double* d_values,
                      const double* d_records, int record_stride, const uint64_t* d_init_bits,
                      const uint64_t* d_prior_bits, const int64_t* d_prior_ref, int64_t* d_rows);
/* d_dest[i] = number of splitter keys (d_splitters[s].key, from sl_select_*) <= row i's key: the rank
 * that owns the row's position in the sorted order. */
SL_API int  sl_adaptive_dest(sl_ctx* ctx, int64_t count, const int64_t* d_rows,
                      const sl_select_state* d_splitters, int nsplit, uint8_t* d_dest);
/* keys[i] = vbits of row i, vals[i] = i: the input of sl_sort_pairs. */
SL_API int  sl_adaptive_sort_keys(sl_ctx* ctx, int64_t m, const int64_t* d_rows, uint64_t* d_keys,
                           int64_t* d_vals);
/* Judge the batches of the m rows at sorted positions [pos0, pos0 + m) (pos0 a multiple of batch;
 * d_order[q] = row of position pos0 + q): d_info[b] = {passes, bound, stop, refine_bound} as in
 * lyapunov.py:537-582, *d_first_break = smallest global batch index that ends the loop (INT64_MAX:
 * none; MIN it over the ranks). */
SL_API int  sl_adaptive_analyse(sl_ctx* ctx, int64_t m, int64_t pos0, int64_t batch,
                         const int64_t* d_rows, const int64_t* d_order, double tau,
                         double safety_factor, int64_t max_refinement, int32_t* d_info,
                         int64_t* d_first_break);
/* d_out_rows[row] = {flat index, safe, refinement} after the loop ended in batch b_star. */
SL_API int  sl_adaptive_apply(sl_ctx* ctx, int64_t m, int64_t pos0, int64_t batch,
                       const int64_t* d_rows, const int64_t* d_order, const int32_t* d_info,
                       double tau, double safety_factor, int64_t b_star, int64_t* d_out_rows);
/* {index, safe, refinement} rows of cells of [lo,hi) -> this range's mask words and refinement
 * array; the initial set is kept (lyapunov.py:601-606); *d_safe_count = bits set. */
SL_API int  sl_adaptive_scatter(sl_ctx* ctx, int64_t lo, int64_t hi, int64_t m,
                         const int64_t* d_out_rows, const uint64_t* d_init_bits,
                         uint64_t* d_safe_bits, int64_t* d_refinement, int64_t* d_safe_count);

/* ---- get_safe_sample / perturb_actions (lyapunov.py:609-797): the glue between the safe set and
 * the point evaluations (sl_eval_points) ------------------------------------------------------------ */
/* d_states[i] = index_to_state(d_indices[i]) on the model's grid (functions.py:714-731). */
SL_API int  sl_index_to_state(sl_ctx* ctx, int64_t count, const int64_t* d_indices, double* d_states);
/* Row i*nperturb + k = [state_i, clip(action_i + perturbation_k, limits)] (lyapunov.py:634-647);
 * d_limits [m][2] may be NULL (no clipping). */
SL_API int  sl_perturb_pairs(sl_ctx* ctx, int64_t count, int d, int m, const double* d_states,
                      const double* d_actions, int nperturb, const double* d_perturbations,
                      const double* d_limits, double* d_pairs);
/* utilities.unique_rows (:496-516) orders rows by memcmp of their raw bytes: d_keys[q] = byte-swapped
 * word `column` of row d_order[q] (NULL: q) - sort by it with sl_sort_pairs, last column first. */
SL_API int  sl_rows_sort_key(sl_ctx* ctx, int64_t count, int words, int column, const int64_t* d_rows,
                      const int64_t* d_order, uint64_t* d_keys);
/* d_flags[q] = 1 when row d_order[q] equals row d_order[q-1] bit for bit (drop it). */
SL_API int  sl_rows_duplicate_flags(sl_ctx* ctx, int64_t count, int words, const int64_t* d_rows,
                             const int64_t* d_order, uint8_t* d_flags);
/* bound_i = sum_j std_ij and inside_i = V(mean_i) + sum_j L_v(mean_i)_j std_ij < c_max
 * (lyapunov.py:716-726); lv_cols = 1 or d. */
SL_API int  sl_sample_bounds(sl_ctx* ctx, int64_t count, int d, int lv_cols, const double* d_std,
                      const double* d_lv, const double* d_value, double c_max, double* d_bound,
                      uint8_t* d_inside);
/* d_inout[i] &= safe_set[state_to_index(point_i)] (lyapunov.py:762-766, functions.py:733-752);
 * d_safe_bits = the mask words of the WHOLE grid. */
SL_API int  sl_state_membership(sl_ctx* ctx, int64_t count, const double* d_points,
                         const uint64_t* d_safe_bits, uint8_t* d_inout);
/* d_out[0] = first index of the largest d_values[i] among rows with d_mask[i] != 0 (NULL: all), NaN
 * counting as largest (np.argmax, lyapunov.py:783, 789), -1 if there is none; d_out[1] = rows seen. */
SL_API int  sl_argmax_masked(sl_ctx* ctx, int64_t count, const double* d_values, const uint8_t* d_mask,
                      int64_t* d_out);
/* discrete_policy_optimization with a safety constraint (reinforcement_learning.py:266-278): d_best[i]
 * = np.argmax_a of d_q[i][a] with the actions that are not allowed at vertex i counted as -inf (first
 * maximiser wins, NaN largest, nothing allowed -> 0).  d_allowed_bits[a * words_per_action + (i >> 6)]
 * bit (i & 63): action a may be taken at vertex i - e.g. the `negative` words of one sl_lyap_sweep
 * per action with that action as the policy (NULL: no constraint).  The [count, A] table of action
 * values stays on the device. */
SL_API int  sl_argmax_rows_masked(sl_ctx* ctx, int64_t count, int n_actions, const double* d_q,
                           const uint64_t* d_allowed_bits, int64_t words_per_action, int32_t* d_best);

/* get_lyapunov_region (lyapunov.py:59-139): the region around the node `start` (flat index on the
 * model's grid) that the reference's priority-queue flood of the function values visits before it
 * reaches the grid boundary or has to descend - computed as a minimax-distance fixpoint.
 *   d_values [nindex] the function on the grid (sl_values), d_work [nindex] scratch (the distances),
 *   d_region [nindex] out: 1 inside; *sweeps_out (may be NULL) relaxation passes used. */
SL_API int  sl_lyapunov_region(sl_ctx* ctx, const double* d_values, int64_t start, double* d_work,
                        uint8_t* d_region, int* sweeps_out);

/* Bit mask <-> byte mask helpers for the bool[N] safe_set of the reference (lyapunov.py:187). */
SL_API int  sl_bits_to_bytes(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, uint8_t* d_bytes);
SL_API int  sl_bytes_to_bits(sl_ctx* ctx, int64_t n, const uint8_t* d_bytes, uint64_t* d_bits);
/* np.where(safe_set) (lyapunov.py:729, the first step of get_safe_sample): the flat indices of the set
 * bits of a mask of n cells, ascending.  Two calls, because the caller sizes d_indices by the count:
 *   sl_bits_count      d_block_counts [ceil(ceil(n/64)/256)] and d_offsets [that + 1] are scratch it fills
 *                      (set bits per 16 384 cells, their exclusive prefix sums); *total_out (HOST) = number
 *                      of set bits - one 8-byte copy and a stream synchronisation;
 *   sl_bits_to_indices d_indices [total] out, with the d_offsets of the count call.
 * Bits of the last word beyond n are ignored. */
SL_API int  sl_bits_count(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, uint32_t* d_block_counts,
                          int64_t* d_offsets, int64_t* total_out);
SL_API int  sl_bits_to_indices(sl_ctx* ctx, int64_t n, const uint64_t* d_bits, const int64_t* d_offsets,
                               int64_t* d_indices);

/* ---- dynamic programming (reinforcement_learning.py:65-140, 213-279) --------------------- */
/* One Jacobi sweep over vertices [lo,hi) of the value grid (auxiliary grid #0):
 *   q(i,a) = r(x_i,u_a) + gamma * V_old(mean f(x_i,u_a))
 * n_actions == 0: u = policy(x_i) (value_iteration, :135-140).
 * n_actions  > 0: u_a = h_actions[a] (row-major [n_actions][m]); writes max_a q and the first
 *                 arg-max (discrete_policy_optimization, :266-279); d_q (may be NULL) gets all q.
 *   d_v_new [hi-lo], d_argmax [hi-lo] (may be NULL), d_stats[2] = {max_i |v_new_i - table_i|,
 *   sum_i (v_new_i - V(x_i))^2} with V(x_i) interpolated as in reinforcement_learning.py:130-133;
 *   the sum (the Bellman error of the current policy, :116-133) is produced for n_actions == 0
 *   only and is 0 otherwise. */
SL_API int  sl_bellman_sweep(sl_ctx* ctx, int64_t lo, int64_t hi, int n_actions, const double* h_actions,
                      double* d_v_new, int32_t* d_argmax, double* d_q, double* d_stats);

/* Successor cache of sl_bellman_sweep.  The next state of (vertex, action) - and where it falls in the
 * value grid: rectangle, unit-cell simplex, barycentric weights - does not depend on the value table
 * (reinforcement_learning.py:89-104; the table enters at :101 only), and a value-iteration loop
 * repeats the sweep with the same dynamics and action set.  The first max sweep (n_actions > 0) over
 * a range therefore keeps the located successors on the device (8 d + 5 bytes per pair: 6.2 GB for
 * 64^4 vertices x 9 actions); later max sweeps over the same range with the same h_actions - and
 * policy-evaluation sweeps (n_actions == 0) whose table policy takes one of those actions at every
 * vertex - gather and combine from it.  Their results are bit-identical to the uncached kernels'.
 * The cache is dropped by whatever changes a successor: sl_model_set with another grid / dynamics
 * description, sl_gp_set_head[_kernel], sl_gp_append_point, sl_gp_configure with another head count,
 * sl_tri_set(slot 0); sl_tri_set_table, the reward, gamma and the policy may change freely.
 *   max_bytes < 0: default budget (a quarter of the device's memory); 0: no cache (frees it);
 *   otherwise the largest allocation allowed - a sweep whose cache would not fit recomputes. */
SL_API int  sl_successor_cache_configure(sl_ctx* ctx, int64_t max_bytes);
typedef struct sl_successor_cache_stats {
    int64_t bytes, max_bytes;     /* allocated / allowed                                         */
    int64_t lo, hi;               /* vertex range of the cached sweep                            */
    int32_t valid, n_actions;     /* 1: the next matching sweep is served from the cache         */
    int64_t fills, hits, policy_hits;   /* sweeps that filled it / max sweeps / policy sweeps served */
} sl_successor_cache_stats;
SL_API int  sl_successor_cache_info(sl_ctx* ctx, sl_successor_cache_stats* out);

/* ---- evaluation at arbitrary points (Function.__call__, lyapunov.py:265-288, 324-376) ------ */
enum sl_eval_what {
    SL_EVAL_VALUE = 1,      /* V(x)                    out [n][1]                  */
    SL_EVAL_POLICY = 2,     /* policy(x)               out [n][m]                  */
    SL_EVAL_DYNAMICS = 3,   /* per-point record        out [n][2+2d] =             */
    SL_EVAL_DECREASE = 4,   /*   [v_decrease_bound, threshold, mean f[d], error[d]] (both selectors) */
    SL_EVAL_LV = 5          /* L_v(x)                  out [n][lv_cols]            */
};
SL_API int  sl_eval_points(sl_ctx* ctx, int what, int64_t n, const double* d_points /* [n][d] */,
                    double* d_out);

/* ---- multi-GPU collectives directly on RCCL ----------------------------- *
 * For callers without torch.distributed (the Python package issues the same exchanges through
 * torch.distributed, backend "nccl" = RCCL).  One communicator per context, one rank per GPU; every
 * call is asynchronous on the context's stream.  RCCL is bound at run time: on a system without
 * librccl.so the calls return SL_ERR_UNSUPPORTED.
 *   sl_comm_unique_id: rank 0 creates the 128-byte id and hands it to the other ranks out of band.
 *   sl_allreduce_result: the per-shard record of sl_lyap_sweep / sl_lyap_finalize becomes the record
 *     of the whole grid IN PLACE: lexicographic min of `fail`, lexicographic max of `last_safe`
 *     and `max_key`, sums of the counters (the reductions of lyapunov.py:512-606 over shards).
 *   sl_allgather: value-table shards of equal size after a Bellman sweep
 *     (reinforcement_learning.py:135-140); sl_allreduce_max_f64: its residual;
 *   sl_allreduce_sum_u64: the radix-select histograms of sl_select_pass.                        */
#define SL_COMM_ID_BYTES 128
SL_API int  sl_comm_unique_id(unsigned char* id_out /* [SL_COMM_ID_BYTES] */);
SL_API int  sl_comm_init(sl_ctx* ctx, const unsigned char* id /* [SL_COMM_ID_BYTES] */, int rank, int world);
SL_API int  sl_comm_destroy(sl_ctx* ctx);
SL_API int  sl_allreduce_result(sl_ctx* ctx, sl_sweep_result* d_result);
SL_API int  sl_allgather(sl_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank);
SL_API int  sl_allreduce_sum_u64(sl_ctx* ctx, uint64_t* d_values, int64_t count);
SL_API int  sl_allreduce_max_f64(sl_ctx* ctx, double* d_values, int64_t count);

/* ---- diagnostics -------------------------------------------------------------------------- */
/* D = A(16x4) * B(4x16) through v_mfma_f64_16x16x4_f64 with this library's fragment maps. */
SL_API int  sl_debug_mfma(sl_ctx* ctx, const double* h_a, const double* h_b, double* h_d);
/* v_mfma_f64_4x4x4_4b_f64 on per-lane operands (nwaves x 64 values each); mode 0: plain, 1..4:
   cbsz = 2, abid = mode - 1 (not a block broadcast for FP64).  Pins the fragment layout in the
   tests. */
SL_API int  sl_debug_mfma4(sl_ctx* ctx, int nwaves, const double* h_a, const double* h_b,
                    const double* h_c, int mode, double* h_d);
/* Sustained FP64 rate probes: which = 0 MFMA, 1 VALU FMA, 2 both interleaved.
 * h_out[3] = {TFLOP/s, sustained shader clock in MHz, shader cycles per MFMA slot per SIMD}. */
SL_API int  sl_debug_fp64_rate(sl_ctx* ctx, int which, int iters, double* h_out);
/* The scaled training inputs X / lengthscales of an uploaded GP head as the kernels read them,
 * h_xs [p][n] (row q = input dimension q): lets the tests compare an incrementally extended head
 * (sl_gp_append_point) with a fresh upload bit for bit. */
SL_API int  sl_debug_gp_inputs(sl_ctx* ctx, int head, double* h_xs);

#ifdef __cplusplus
}
#endif
#endif /* SL_HIP_H */
