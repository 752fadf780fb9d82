// sl_policy_net.hip - a NeuralNetwork as the policy (safe_learning/functions.py:1663-1729; the policy
// of examples/inverted_pendulum.ipynb:215 - `NeuralNetwork(layers=[32, 32, 1], nonlinearities=[relu,
// relu, tanh], output_scale=...)` - and of the reinforcement-learning notebooks).
//
// The reference's network is a chain of tf.layers.dense: net <- act_l(net W_l + b_l) for every entry
// of `layers` (units per layer; the input width is the state dimension), the last layer without a
// bias, the output multiplied by output_scale (functions.py:1702-1729).
//
// The sweep kernels evaluate closed-form and table policies in registers; a network's activations
// (up to 64 per layer) have no room there.  The same answer as for an interpolated policy in front of
// k_gp_sweep4 (sl_kernels.hip: k_policy_table): the policy is evaluated ONCE per cell of the call
// into a per-cell action table (8 m bytes per cell) by k_policy_network, and the sweep / point
// evaluation runs with SL_POLICY_TABLE on that table - every kernel of the library takes a network
// policy that way.  SlPolicyTableScope (sl_common.h) brackets an entry point with the swap.
#include "sl_common.h"

namespace {

__device__ __forceinline__ double pnet_act(int a, double x) {
    if (a == 1) return sl_tanh(x);                 // (sl_model.h: the device routine of the network activations)
    if (a == 2) return x > 0.0 ? x : 0.0;
    if (a == 3) return 1.0 / (1.0 + exp(-x));
    return x;
}

}  // namespace

// One thread per cell (or explicit point).  The weights are read with wave-uniform addresses
// (scalar loads); the activations of a layer live in per-lane arrays.
__global__ __launch_bounds__(SL_BLOCK) void k_policy_network(const SlDevModel M, const SlPolicyNet net,
                                                             int64_t lo, int64_t hi,
                                                             const double* __restrict__ points,
                                                             double* __restrict__ actions) {
    const int d = M.m.grid.d, m = net.dims[net.nlayers];
    for (int64_t idx = lo + (int64_t)blockIdx.x * SL_BLOCK + threadIdx.x; idx < hi;
         idx += (int64_t)gridDim.x * SL_BLOCK) {
        double a[SL_NN_MAXW], b[SL_NN_MAXW], x[SL_P];
        sl_cell_state(M, d, idx, points, x);
        for (int k = 0; k < d; ++k) a[k] = x[k];
        for (int l = 0; l < net.nlayers; ++l) {
            const int in = net.dims[l], out = net.dims[l + 1];
            const double* K = net.params + net.koff[l];            // [in][out]: y = x K (+ bias)
            const double* bias = net.boff[l] >= 0 ? net.params + net.boff[l] : nullptr;
            for (int o = 0; o < out; ++o) {
                double s = 0.0;
                for (int i = 0; i < in; ++i) s = fma(a[i], K[i * out + o], s);
                if (bias) s = s + bias[o];
                b[o] = pnet_act(net.act[l], s);
            }
            for (int o = 0; o < out; ++o) a[o] = b[o];
        }
        for (int o = 0; o < m; ++o) actions[(idx - lo) * m + o] = a[o] * net.scale;
    }
}

extern "C" int sl_policy_network_set(sl_ctx* ctx, int nlayers, const int32_t* h_dims,
                                     const int32_t* h_activations, const double* h_kernels,
                                     const double* h_biases, const int32_t* h_has_bias,
                                     double output_scale) {
    if (!ctx || !h_dims || !h_activations || !h_kernels)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: NULL argument");
    if (nlayers < 1 || nlayers > SL_MAX_NN_LAYERS)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: %d layers (max %d)", nlayers,
                       SL_MAX_NN_LAYERS);
    SlPolicyNet n;
    memset(&n, 0, sizeof(n));
    n.nlayers = nlayers;
    n.scale = output_scale;
    int64_t total = 0;
    for (int l = 0; l <= nlayers; ++l) {
        if (h_dims[l] < 1 || h_dims[l] > SL_NN_MAXW)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: width %d outside [1,%d]",
                           h_dims[l], SL_NN_MAXW);
        n.dims[l] = h_dims[l];
    }
    if (h_dims[nlayers] > SL_MAX_ACTION_DIM)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: %d outputs (max %d action dimensions)",
                       h_dims[nlayers], SL_MAX_ACTION_DIM);
    std::vector<double> params;
    const double* kp = h_kernels;
    const double* bp = h_biases;
    for (int l = 0; l < nlayers; ++l) {
        if (h_activations[l] < 0 || h_activations[l] > 3)
            return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: activation code %d", h_activations[l]);
        n.act[l] = h_activations[l];
        const int64_t count = (int64_t)h_dims[l] * h_dims[l + 1];
        n.koff[l] = (int32_t)params.size();
        params.insert(params.end(), kp, kp + count);
        kp += count;
        n.boff[l] = -1;
        if (h_has_bias && h_has_bias[l]) {
            if (!h_biases) return sl_fail(ctx, SL_ERR_INVALID, "sl_policy_network_set: biases missing");
            n.boff[l] = (int32_t)params.size();
            params.insert(params.end(), bp, bp + h_dims[l + 1]);
            bp += h_dims[l + 1];
        }
        total += count;
    }
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));       // kernels may still read the old parameters
    if (ctx->d_pnet_params) { (void)hipFree(ctx->d_pnet_params); ctx->d_pnet_params = nullptr; }
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_pnet_params, sizeof(double) * params.size()));
    SL_HIP_CHECK(ctx, hipMemcpy(ctx->d_pnet_params, params.data(), sizeof(double) * params.size(),
                                hipMemcpyHostToDevice));
    n.params = ctx->d_pnet_params;
    n.set = 1;
    ctx->pnet = n;
    ++ctx->policy_token;                         // what the sweeps derive from the policy alone is stale
    return SL_OK;
}

// ---- the swap around an entry point ------------------------------------------------------------
SlPolicyTableScope::SlPolicyTableScope(sl_ctx* c, int64_t lo, int64_t hi, const double* d_points)
    : ctx(c), swapped(false), rc(SL_OK) {
    if (!ctx || !ctx->model_set || ctx->h_model.m.policy.kind != SL_POLICY_NETWORK) return;
    const SlDevModel& M = ctx->h_model;
    const SlPolicyNet& net = ctx->pnet;
    if (!net.set) { rc = sl_fail(ctx, SL_ERR_INVALID, "network policy not set (sl_policy_network_set)"); return; }
    const int d = M.m.grid.d, m = M.m.policy.m;
    if (net.dims[0] != d || net.dims[net.nlayers] != m) {
        rc = sl_fail(ctx, SL_ERR_INVALID, "network policy maps %d -> %d, the model needs %d -> %d",
                     net.dims[0], net.dims[net.nlayers], d, m);
        return;
    }
    if (hi <= lo) return;
    const size_t need = sizeof(double) * (size_t)(hi - lo) * m;
    if (need > ctx->policy_actions_bytes) {
        if (hipSetDevice(ctx->device) != hipSuccess) { rc = sl_fail(ctx, SL_ERR_HIP, "hipSetDevice failed"); return; }
        if (ctx->d_policy_actions) (void)hipFree(ctx->d_policy_actions);
        ctx->d_policy_actions = nullptr;
        ctx->policy_actions_bytes = 0;
        if (hipMalloc(&ctx->d_policy_actions, need) != hipSuccess) {
            (void)hipGetLastError();
            rc = sl_fail(ctx, SL_ERR_NOMEM, "network policy: %zu bytes for the action table", need);
            return;
        }
        ctx->policy_actions_bytes = need;
    }
    double* act = reinterpret_cast<double*>(ctx->d_policy_actions);
    hipLaunchKernelGGL(k_policy_network, dim3(sl_grid_blocks(hi - lo)), dim3(SL_BLOCK), 0, ctx->stream, M, net,
                       lo, hi, d_points, act);
    if (hipGetLastError() != hipSuccess) { rc = sl_fail(ctx, SL_ERR_HIP, "k_policy_network launch failed"); return; }
    saved = ctx->h_model.m.policy;
    ctx->h_model.m.policy.kind = SL_POLICY_TABLE;
    ctx->h_model.m.policy.d_table = act - lo * m;        // indexed by the cell (or point) number
    swapped = true;
}

SlPolicyTableScope::~SlPolicyTableScope() {
    if (swapped) ctx->h_model.m.policy = saved;
}
