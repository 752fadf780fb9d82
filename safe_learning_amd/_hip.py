"""ctypes binding of libslhip.so (``include/sl_hip.h``) - the only way into the HIP engine.

There is no CPU fallback: if the shared library is missing or no GPU is present the loader /
context raise.  Device memory is owned by torch tensors (plumbing only); this module passes
their ``data_ptr()`` across the C ABI.
"""

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SL_LIB_PATH: another build of the same library (kernel A/B runs, tools/build_variant.sh)
LIB_PATH = os.environ.get("SL_LIB_PATH") or os.path.join(HERE, "libslhip.so")

MAX_STATE_DIM = 6
MAX_ACTION_DIM = 2
MAX_INPUT_DIM = 8
MAX_GP_HEADS = 6
MAX_NN_LAYERS = 4
MAX_SIMPLICES = 32

# enum values of include/sl_hip.h
POLICY_LINEAR, POLICY_CONST, POLICY_TABLE, POLICY_TRI, POLICY_NETWORK = 1, 2, 3, 4, 5
DYN_LINEAR, DYN_PENDULUM, DYN_CARTPOLE, DYN_GP = 1, 2, 3, 4
V_QUADRATIC, V_TRI, V_NETWORK = 1, 2, 3
LIP_CONST, LIP_ABS_LINEAR, LIP_NORM_LINEAR, LIP_ABS_GRAD, LIP_NORM_GRAD = 0, 1, 2, 3, 4
LF_CONST, LF_AFFINE_NORM1 = 0, 1
EVAL_VALUE, EVAL_POLICY, EVAL_DYNAMICS, EVAL_DECREASE, EVAL_LV = 1, 2, 3, 4, 5

c_double_p = C.POINTER(C.c_double)


class GridDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("reserved", C.c_int32),
                ("num_points", C.c_int64 * MAX_STATE_DIM),
                ("offset", C.c_double * MAX_STATE_DIM),
                ("unit_maxes", C.c_double * MAX_STATE_DIM),
                ("upper", C.c_double * MAX_STATE_DIM)]


class PolicyDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m", C.c_int32), ("saturate", C.c_int32),
                ("reserved", C.c_int32),
                ("matrix", (C.c_double * MAX_STATE_DIM) * MAX_ACTION_DIM),
                ("lower", C.c_double * MAX_ACTION_DIM), ("upper", C.c_double * MAX_ACTION_DIM),
                ("constant", C.c_double * MAX_ACTION_DIM),
                ("d_table", C.c_void_p)]


class DynamicsDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("normalize", C.c_int32),
                ("matrix", (C.c_double * MAX_INPUT_DIM) * MAX_STATE_DIM),
                ("tx", C.c_double * MAX_STATE_DIM), ("tx_inv", C.c_double * MAX_STATE_DIM),
                ("tu", C.c_double * MAX_ACTION_DIM),
                ("coef", C.c_double * 16)]


class ValueDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("negate", C.c_int32),
                ("matrix", (C.c_double * MAX_INPUT_DIM) * MAX_INPUT_DIM)]


class LipschitzDesc(C.Structure):
    _fields_ = [("lv_kind", C.c_int32), ("lv_cols", C.c_int32), ("lv_const", C.c_double),
                ("lv_matrix", (C.c_double * MAX_STATE_DIM) * MAX_STATE_DIM),
                ("lf_const", C.c_double), ("tau", C.c_double),
                ("lf_kind", C.c_int32), ("lf_reserved", C.c_int32),
                ("lf_matrix", (C.c_double * MAX_STATE_DIM) * MAX_STATE_DIM)]


KERNEL_RBF, KERNEL_MATERN32, KERNEL_LINEAR = 0, 1, 2
KERNEL_MAX_FACTORS = 8


class GpKernelFactor(C.Structure):
    _fields_ = [("kind", C.c_int32), ("product", C.c_int32),
                ("variance", C.c_double * MAX_INPUT_DIM),
                ("inv_lengthscales", C.c_double * MAX_INPUT_DIM)]


class GpKernel(C.Structure):
    """``sl_gp_kernel``: a sum of products of RBF / Matern32 / Linear leaves."""
    _fields_ = [("nfactors", C.c_int32), ("reserved", C.c_int32),
                ("factor", GpKernelFactor * KERNEL_MAX_FACTORS)]


class ModelDesc(C.Structure):
    _fields_ = [("grid", GridDesc), ("policy", PolicyDesc), ("dynamics", DynamicsDesc),
                ("value", ValueDesc), ("lipschitz", LipschitzDesc), ("reward", ValueDesc),
                ("gamma", C.c_double)]


class Key(C.Structure):
    _fields_ = [("vbits", C.c_uint64), ("index", C.c_int64)]


class SuccessorCacheStats(C.Structure):
    """``sl_successor_cache_stats`` (include/sl_hip.h)."""
    _fields_ = [("bytes", C.c_int64), ("max_bytes", C.c_int64), ("lo", C.c_int64), ("hi", C.c_int64),
                ("valid", C.c_int32), ("n_actions", C.c_int32), ("fills", C.c_int64),
                ("hits", C.c_int64), ("policy_hits", C.c_int64)]


# sl_sweep_result as int64 words (a torch int64[8] tensor backs it on the device)
RESULT_WORDS = 8
R_FAIL_V, R_FAIL_I, R_LAST_V, R_LAST_I, R_MAX_V, R_MAX_I, R_BELOW, R_SAFE = range(8)
SORT_COUNT_WORDS = 256 * 2048          # uint32 scratch of sl_sort_pairs / sl_partition_by_digit
ADAPTIVE_ROW_WORDS = 6                 # vbits, index, decrease, threshold(tau = 1), refinement, flags
# sl_select_state as int64 words: prefix, remaining, key.vbits, key.index, rank, none, pad, pad
SELECT_WORDS = 8
S_PREFIX, S_REMAINING, S_KEY_V, S_KEY_I, S_RANK, S_NONE = range(6)

EXPORTS = [
    "sl_version", "sl_ctx_create", "sl_ctx_destroy", "sl_last_error", "sl_ctx_synchronize",
    "sl_last_kernel",
    "sl_model_set", "sl_gp_set_head", "sl_gp_set_head_kernel", "sl_gp_append_point", "sl_gp_configure", "sl_tri_set", "sl_tri_set_table",
    "sl_network_set", "sl_policy_network_set", "sl_values", "sl_lyap_sweep", "sl_lyap_finalize", "sl_select_pass",
    "sl_values_implicit", "sl_fold_results", "sl_lyap_finalize_dev", "sl_refinement_carry", "sl_select_begin",
    "sl_select_hist", "sl_select_digit",
    "sl_sort_pairs", "sl_partition_by_digit", "sl_gather_rows", "sl_adaptive_pack", "sl_adaptive_dest",
    "sl_adaptive_sort_keys", "sl_adaptive_analyse", "sl_adaptive_apply", "sl_adaptive_scatter",
    "sl_index_to_state", "sl_perturb_pairs", "sl_rows_sort_key", "sl_rows_duplicate_flags",
    "sl_sample_bounds", "sl_state_membership", "sl_argmax_masked", "sl_argmax_rows_masked", "sl_lyapunov_region",
    "sl_bits_to_bytes", "sl_bytes_to_bits", "sl_bits_count", "sl_bits_to_indices", "sl_bellman_sweep", "sl_successor_cache_configure",
    "sl_successor_cache_info", "sl_eval_points", "sl_timing_configure", "sl_timing_collect",
    "sl_comm_unique_id", "sl_comm_init", "sl_comm_destroy", "sl_allreduce_result", "sl_allgather",
    "sl_allreduce_sum_u64", "sl_allreduce_max_f64",
    "sl_debug_mfma", "sl_debug_mfma4", "sl_debug_fp64_rate", "sl_debug_gp_inputs",
]

_lib = None


class HipEngineError(RuntimeError):
    """Raised for every failure of the HIP engine (no silent fallbacks)."""


def load_library():
    """Load libslhip.so (built by ``python -m safe_learning_amd._build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipEngineError(
            "libslhip.so is missing (%s). Build it with `python -m safe_learning_amd._build`; "
            "safe_learning_amd has no CPU fallback." % LIB_PATH)
    # torch first: libslhip.so must bind the HIP runtime that torch ships (loading the system
    # libamdhip64 before torch's copy leaves the process with two runtimes, one without devices)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.sl_version.restype = C.c_int
    lib.sl_last_error.restype = C.c_char_p
    lib.sl_last_error.argtypes = [C.c_void_p]
    dev_early = bool(os.environ.get("SL_LIB_PATH"))
    lib.sl_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.sl_ctx_destroy.argtypes = [C.c_void_p]
    lib.sl_ctx_synchronize.argtypes = [C.c_void_p]
    lib.sl_model_set.argtypes = [C.c_void_p, C.POINTER(ModelDesc)]
    lib.sl_gp_set_head.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   c_double_p, c_double_p, c_double_p, C.c_double, c_double_p]
    lib.sl_gp_set_head_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          c_double_p, c_double_p, c_double_p, C.POINTER(GpKernel)]
    lib.sl_gp_append_point.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]
    lib.sl_gp_configure.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.sl_tri_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(GridDesc), C.c_int,
                               C.POINTER(C.c_int32), c_double_p, c_double_p, C.c_int, C.c_int,
                               C.c_void_p]
    lib.sl_tri_set_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.sl_network_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), c_double_p]
    if not dev_early or hasattr(lib, "sl_policy_network_set"):
        lib.sl_policy_network_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              c_double_p, c_double_p, C.POINTER(C.c_int32), C.c_double]
    lib.sl_values.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.sl_lyap_sweep.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_lyap_finalize.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, Key, Key, C.c_void_p, C.c_void_p]
    lib.sl_select_pass.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                   C.c_uint64, C.c_uint64, C.c_void_p]
    lib.sl_values_implicit.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.sl_fold_results.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.sl_lyap_finalize_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_refinement_carry.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_select_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
    lib.sl_select_hist.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]
    lib.sl_select_digit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    vp, i64 = C.c_void_p, C.c_int64
    lib.sl_sort_pairs.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.sl_partition_by_digit.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.sl_gather_rows.argtypes = [vp, i64, C.c_int, vp, vp, vp]
    lib.sl_adaptive_pack.argtypes = [vp, i64, i64, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.sl_adaptive_dest.argtypes = [vp, i64, vp, vp, C.c_int, vp]
    lib.sl_adaptive_sort_keys.argtypes = [vp, i64, vp, vp, vp]
    lib.sl_adaptive_analyse.argtypes = [vp, i64, i64, i64, vp, vp, C.c_double, C.c_double, i64, vp, vp]
    lib.sl_adaptive_apply.argtypes = [vp, i64, i64, i64, vp, vp, vp, C.c_double, C.c_double, i64, vp]
    lib.sl_adaptive_scatter.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.sl_index_to_state.argtypes = [vp, i64, vp, vp]
    lib.sl_perturb_pairs.argtypes = [vp, i64, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]
    lib.sl_rows_sort_key.argtypes = [vp, i64, C.c_int, C.c_int, vp, vp, vp]
    lib.sl_rows_duplicate_flags.argtypes = [vp, i64, C.c_int, vp, vp, vp]
    lib.sl_sample_bounds.argtypes = [vp, i64, C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, vp]
    lib.sl_state_membership.argtypes = [vp, i64, vp, vp, vp]
    lib.sl_argmax_masked.argtypes = [vp, i64, vp, vp, vp]
    lib.sl_argmax_rows_masked.argtypes = [vp, i64, C.c_int, vp, vp, i64, vp]
    lib.sl_lyapunov_region.argtypes = [vp, vp, i64, vp, vp, C.POINTER(C.c_int)]
    lib.sl_bits_to_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sl_bits_count.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.sl_bits_to_indices.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_bytes_to_bits.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sl_bellman_sweep.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, c_double_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    # (a development library of another revision, SL_LIB_PATH, may lack the newest entry points: A/B
    # runs of the kernels both have; the shipped library is checked symbol by symbol, tests/test_abi.py)
    dev = bool(os.environ.get("SL_LIB_PATH"))
    if not dev or hasattr(lib, "sl_successor_cache_configure"):
        lib.sl_successor_cache_configure.argtypes = [C.c_void_p, C.c_int64]
        lib.sl_successor_cache_info.argtypes = [C.c_void_p, C.POINTER(SuccessorCacheStats)]
    lib.sl_eval_points.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    if not dev or hasattr(lib, "sl_timing_configure"):
        lib.sl_timing_configure.argtypes = [C.c_void_p, C.c_int]
        lib.sl_timing_collect.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int, C.POINTER(C.c_int)]
    lib.sl_comm_unique_id.argtypes = [C.c_char_p]
    lib.sl_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.sl_comm_destroy.argtypes = [C.c_void_p]
    lib.sl_allreduce_result.argtypes = [C.c_void_p, C.c_void_p]
    lib.sl_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.sl_allreduce_sum_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sl_allreduce_max_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sl_debug_mfma.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
    lib.sl_debug_fp64_rate.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p]
    lib.sl_debug_gp_inputs.argtypes = [C.c_void_p, C.c_int, c_double_p]
    lib.sl_debug_mfma4.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int,
                                   c_double_p]
    for name in EXPORTS:
        if name not in ("sl_last_error", "sl_last_kernel") and (not dev or hasattr(lib, name)):
            getattr(lib, name).restype = C.c_int
    lib.sl_last_error.restype = C.c_char_p
    lib.sl_last_kernel.restype = C.c_char_p
    lib.sl_last_kernel.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _as_c(array):
    array = np.ascontiguousarray(array, dtype=np.float64)
    return array, array.ctypes.data_as(c_double_p)


def _ptr(tensor):
    """Device pointer of a torch tensor (or None)."""
    if tensor is None:
        return None
    return C.c_void_p(tensor.data_ptr())


class Context(object):
    """One engine context = one GPU + one HIP stream (torch's current stream)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise HipEngineError("no GPU visible to torch; safe_learning_amd has no CPU fallback")
        self.lib = load_library()
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
        handle = C.c_void_p()
        rc = self.lib.sl_ctx_create(self.device, C.c_void_p(stream), C.byref(handle))
        if rc != 0:
            raise HipEngineError("sl_ctx_create failed: %s" % self.lib.sl_last_error(None).decode())
        self.handle = handle
        self._keepalive = {}        # slot -> the device table the engine currently points at

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sl_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.sl_last_error(self.handle)
            raise HipEngineError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    # ---- model ---------------------------------------------------------------------------
    def model_set(self, desc):
        self.check(self.lib.sl_model_set(self.handle, C.byref(desc)), "sl_model_set")

    def gp_set_head(self, head, X, Linv, alpha, col0, variance, lengthscales):
        X, pX = _as_c(X)
        Linv, pL = _as_c(Linv)
        alpha, pA = _as_c(alpha)
        ls, pls = _as_c(lengthscales)
        n, p = X.shape
        self.check(self.lib.sl_gp_set_head(self.handle, head, n, p, alpha.shape[1], col0, pX, pL,
                                           pA, float(variance), pls), "sl_gp_set_head")

    def gp_set_head_kernel(self, head, X, Linv, alpha, col0, factors):
        """``factors``: ``[(kind, product, variance[p], inv_lengthscales[p])]`` - a sum of products
        of leaf kernels (``sl_gp_kernel`` of include/sl_hip.h)."""
        X, pX = _as_c(X)
        Linv, pL = _as_c(Linv)
        alpha, pA = _as_c(alpha)
        n, p = X.shape
        if len(factors) > KERNEL_MAX_FACTORS:
            raise ValueError("kernel with %d leaf factors (engine limit %d)" % (len(factors), KERNEL_MAX_FACTORS))
        spec = GpKernel()
        spec.nfactors = len(factors)
        for f, (kind, product, variance, inv_ls) in enumerate(factors):
            spec.factor[f].kind, spec.factor[f].product = int(kind), int(product)
            for q in range(p):
                spec.factor[f].variance[q] = float(variance[q])
                spec.factor[f].inv_lengthscales[q] = float(inv_ls[q])
        self.check(self.lib.sl_gp_set_head_kernel(self.handle, head, n, p, alpha.shape[1], col0, pX,
                                                  pL, pA, C.byref(spec)), "sl_gp_set_head_kernel")

    def gp_append_point(self, head, x, linv_row, alpha_new):
        """One more training point for an uploaded head; False if the head has to be re-packed."""
        x, px = _as_c(x)
        row, pr = _as_c(linv_row)
        a, pa = _as_c(alpha_new)
        rc = self.lib.sl_gp_append_point(self.handle, head, px, pr, pa)
        if rc == -3:                                   # SL_ERR_UNSUPPORTED: capacity exhausted
            return False
        self.check(rc, "sl_gp_append_point")
        return True

    def gp_inputs(self, head, n, p):
        """Scaled training inputs of an uploaded head as the kernels read them, ``[p, n]``."""
        out = np.empty((p, n), dtype=np.float64)
        self.check(self.lib.sl_debug_gp_inputs(self.handle, head, out.ctypes.data_as(c_double_p)),
                   "sl_debug_gp_inputs")
        return out

    def gp_configure(self, nheads, beta):
        self.check(self.lib.sl_gp_configure(self.handle, nheads, float(beta)), "sl_gp_configure")

    def tri_set(self, slot, grid_desc, simplices, hyperplanes, discrete_points, project, ncols,
                table):
        simplices = np.ascontiguousarray(simplices, dtype=np.int32)
        hyper, ph = _as_c(hyperplanes)
        pts, pp = _as_c(np.concatenate(discrete_points))
        self._keepalive[slot] = table
        self.check(self.lib.sl_tri_set(self.handle, slot, C.byref(grid_desc), len(simplices),
                                       simplices.ctypes.data_as(C.POINTER(C.c_int32)), ph, pp,
                                       int(bool(project)), ncols, _ptr(table)), "sl_tri_set")

    def tri_set_table(self, slot, table):
        self._keepalive[slot] = table
        self.check(self.lib.sl_tri_set_table(self.handle, slot, _ptr(table)), "sl_tri_set_table")

    def network_set(self, dims, activations, kernels):
        dims = np.ascontiguousarray(dims, dtype=np.int32)
        acts = np.ascontiguousarray(activations, dtype=np.int32)
        flat, pk = _as_c(np.concatenate([np.asarray(k, dtype=np.float64).ravel() for k in kernels]))
        self.check(self.lib.sl_network_set(self.handle, len(acts),
                                           dims.ctypes.data_as(C.POINTER(C.c_int32)),
                                           acts.ctypes.data_as(C.POINTER(C.c_int32)), pk),
                   "sl_network_set")

    def policy_network_set(self, dims, activations, kernels, biases, output_scale):
        """``sl_policy_network_set``: ``kernels[l]`` is ``[in, out]``, ``biases[l]`` an ``[out]`` array or None."""
        dims = np.ascontiguousarray(dims, dtype=np.int32)
        acts = np.ascontiguousarray(activations, dtype=np.int32)
        flat, pk = _as_c(np.concatenate([np.asarray(k, dtype=np.float64).ravel() for k in kernels]))
        has = np.ascontiguousarray([0 if b is None else 1 for b in biases], dtype=np.int32)
        present = [np.asarray(b, dtype=np.float64).ravel() for b in biases if b is not None]
        bflat, pb = _as_c(np.concatenate(present) if present else np.zeros(1))
        self.check(self.lib.sl_policy_network_set(self.handle, len(acts),
                                                  dims.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  acts.ctypes.data_as(C.POINTER(C.c_int32)), pk, pb,
                                                  has.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  float(output_scale)), "sl_policy_network_set")

    # ---- passes --------------------------------------------------------------------------
    def values(self, lo, hi, d_values):
        self.check(self.lib.sl_values(self.handle, lo, hi, _ptr(d_values)), "sl_values")

    def lyap_sweep(self, lo, hi, d_init_bits, d_values, d_neg_bits, d_result, d_dbg=None):
        self.check(self.lib.sl_lyap_sweep(self.handle, lo, hi, _ptr(d_init_bits), _ptr(d_values),
                                          _ptr(d_neg_bits), _ptr(d_result), _ptr(d_dbg)),
                   "sl_lyap_sweep")

    def lyap_finalize(self, lo, hi, d_values, d_init_bits, d_prev_bits, key_star, key_keep,
                      d_safe_bits, d_result):
        self.check(self.lib.sl_lyap_finalize(self.handle, lo, hi, _ptr(d_values), _ptr(d_init_bits),
                                             _ptr(d_prev_bits), Key(*key_star), Key(*key_keep),
                                             _ptr(d_safe_bits), _ptr(d_result)), "sl_lyap_finalize")

    def select_pass(self, lo, hi, d_values, which, byte, prefix, vbits_equal, d_hist):
        self.check(self.lib.sl_select_pass(self.handle, lo, hi, _ptr(d_values), which, byte,
                                           C.c_uint64(prefix), C.c_uint64(vbits_equal),
                                           _ptr(d_hist)), "sl_select_pass")

    # ---- the same passes with every decision read from device memory (sl_level.hip) -------
    def values_implicit(self):
        """True when ``d_values=None`` is allowed: quadratic V whose ordering keys the passes
        recompute from the cell index (``sl_values_implicit``)."""
        out = C.c_int(0)
        self.check(self.lib.sl_values_implicit(self.handle, C.byref(out)), "sl_values_implicit")
        return bool(out.value)

    def fold_results(self, d_records, count, d_out):
        self.check(self.lib.sl_fold_results(self.handle, _ptr(d_records), count, _ptr(d_out)),
                   "sl_fold_results")

    def lyap_finalize_dev(self, lo, hi, d_values, d_init_bits, d_prev_bits, d_folded, d_keep,
                          d_safe_bits, d_result):
        self.check(self.lib.sl_lyap_finalize_dev(self.handle, lo, hi, _ptr(d_values),
                                                 _ptr(d_init_bits), _ptr(d_prev_bits), _ptr(d_folded),
                                                 _ptr(d_keep), _ptr(d_safe_bits), _ptr(d_result)),
                   "sl_lyap_finalize_dev")

    def refinement_carry(self, lo, hi, d_values, d_init_bits, d_neg_bits, d_folded, d_keep, d_refinement):
        self.check(self.lib.sl_refinement_carry(self.handle, lo, hi, _ptr(d_values), _ptr(d_init_bits),
                                                _ptr(d_neg_bits), _ptr(d_folded), _ptr(d_keep),
                                                _ptr(d_refinement)), "sl_refinement_carry")

    def select_begin(self, d_state, k, batch, d_folded, n_total):
        self.check(self.lib.sl_select_begin(self.handle, _ptr(d_state), k, batch, _ptr(d_folded),
                                            n_total), "sl_select_begin")

    def select_hist(self, lo, hi, d_values, which, byte, d_state, d_hist):
        self.check(self.lib.sl_select_hist(self.handle, lo, hi, _ptr(d_values), which, byte,
                                           _ptr(d_state), _ptr(d_hist)), "sl_select_hist")

    def select_digit(self, which, byte, d_hist, d_state):
        self.check(self.lib.sl_select_digit(self.handle, which, byte, _ptr(d_hist), _ptr(d_state)),
                   "sl_select_digit")

    # ---- sort / partition / the adaptive branch (sl_adaptive.hip) --------------------------
    def sort_pairs(self, n, d_keys, d_vals, d_keys_tmp, d_vals_tmp, d_counts):
        self.check(self.lib.sl_sort_pairs(self.handle, n, _ptr(d_keys), _ptr(d_vals), _ptr(d_keys_tmp),
                                          _ptr(d_vals_tmp), _ptr(d_counts)), "sl_sort_pairs")

    def partition_by_digit(self, n, d_digits, d_perm, d_bucket_counts, d_counts):
        self.check(self.lib.sl_partition_by_digit(self.handle, n, _ptr(d_digits), _ptr(d_perm),
                                                  _ptr(d_bucket_counts), _ptr(d_counts)),
                   "sl_partition_by_digit")

    def gather_rows(self, count, words, d_perm, d_rows_in, d_rows_out):
        self.check(self.lib.sl_gather_rows(self.handle, count, words, _ptr(d_perm), _ptr(d_rows_in),
                                           _ptr(d_rows_out)), "sl_gather_rows")

    def adaptive_pack(self, lo, hi, d_values, d_records, stride, d_init_bits, d_prior_bits,
                      d_prior_ref, d_rows):
        self.check(self.lib.sl_adaptive_pack(self.handle, lo, hi, _ptr(d_values), _ptr(d_records), stride,
                                             _ptr(d_init_bits), _ptr(d_prior_bits), _ptr(d_prior_ref),
                                             _ptr(d_rows)), "sl_adaptive_pack")

    def adaptive_dest(self, count, d_rows, d_splitters, nsplit, d_dest):
        self.check(self.lib.sl_adaptive_dest(self.handle, count, _ptr(d_rows), _ptr(d_splitters), nsplit,
                                             _ptr(d_dest)), "sl_adaptive_dest")

    def adaptive_sort_keys(self, m, d_rows, d_keys, d_vals):
        self.check(self.lib.sl_adaptive_sort_keys(self.handle, m, _ptr(d_rows), _ptr(d_keys),
                                                  _ptr(d_vals)), "sl_adaptive_sort_keys")

    def adaptive_analyse(self, m, pos0, batch, d_rows, d_order, tau, safety_factor, max_refinement,
                         d_info, d_first_break):
        self.check(self.lib.sl_adaptive_analyse(self.handle, m, pos0, batch, _ptr(d_rows), _ptr(d_order),
                                                float(tau), float(safety_factor), int(max_refinement),
                                                _ptr(d_info), _ptr(d_first_break)), "sl_adaptive_analyse")

    def adaptive_apply(self, m, pos0, batch, d_rows, d_order, d_info, tau, safety_factor, b_star,
                       d_out_rows):
        self.check(self.lib.sl_adaptive_apply(self.handle, m, pos0, batch, _ptr(d_rows), _ptr(d_order),
                                              _ptr(d_info), float(tau), float(safety_factor), int(b_star),
                                              _ptr(d_out_rows)), "sl_adaptive_apply")

    def adaptive_scatter(self, lo, hi, m, d_out_rows, d_init_bits, d_safe_bits, d_refinement,
                         d_safe_count):
        self.check(self.lib.sl_adaptive_scatter(self.handle, lo, hi, m, _ptr(d_out_rows),
                                                _ptr(d_init_bits), _ptr(d_safe_bits), _ptr(d_refinement),
                                                _ptr(d_safe_count)), "sl_adaptive_scatter")

    # ---- get_safe_sample glue (sl_sample.hip) ----------------------------------------------
    def index_to_state(self, count, d_indices, d_states):
        self.check(self.lib.sl_index_to_state(self.handle, count, _ptr(d_indices), _ptr(d_states)),
                   "sl_index_to_state")

    def perturb_pairs(self, count, d, m, d_states, d_actions, nperturb, d_perturbations, d_limits,
                      d_pairs):
        self.check(self.lib.sl_perturb_pairs(self.handle, count, d, m, _ptr(d_states), _ptr(d_actions),
                                             nperturb, _ptr(d_perturbations), _ptr(d_limits),
                                             _ptr(d_pairs)), "sl_perturb_pairs")

    def rows_sort_key(self, count, words, column, d_rows, d_order, d_keys):
        self.check(self.lib.sl_rows_sort_key(self.handle, count, words, column, _ptr(d_rows),
                                             _ptr(d_order), _ptr(d_keys)), "sl_rows_sort_key")

    def rows_duplicate_flags(self, count, words, d_rows, d_order, d_flags):
        self.check(self.lib.sl_rows_duplicate_flags(self.handle, count, words, _ptr(d_rows),
                                                    _ptr(d_order), _ptr(d_flags)),
                   "sl_rows_duplicate_flags")

    def sample_bounds(self, count, d, lv_cols, d_std, d_lv, d_value, c_max, d_bound, d_inside):
        self.check(self.lib.sl_sample_bounds(self.handle, count, d, lv_cols, _ptr(d_std), _ptr(d_lv),
                                             _ptr(d_value), float(c_max), _ptr(d_bound),
                                             _ptr(d_inside)), "sl_sample_bounds")

    def state_membership(self, count, d_points, d_safe_bits, d_inout):
        self.check(self.lib.sl_state_membership(self.handle, count, _ptr(d_points), _ptr(d_safe_bits),
                                                _ptr(d_inout)), "sl_state_membership")

    def argmax_masked(self, count, d_values, d_mask, d_out):
        self.check(self.lib.sl_argmax_masked(self.handle, count, _ptr(d_values), _ptr(d_mask),
                                             _ptr(d_out)), "sl_argmax_masked")

    def argmax_rows_masked(self, count, n_actions, d_q, d_allowed_bits, words_per_action, d_best):
        self.check(self.lib.sl_argmax_rows_masked(self.handle, count, n_actions, _ptr(d_q),
                                                  _ptr(d_allowed_bits), words_per_action, _ptr(d_best)),
                   "sl_argmax_rows_masked")

    def lyapunov_region(self, d_values, start, d_work, d_region):
        """-> relaxation passes used (``sl_lyapunov_region``)."""
        sweeps = C.c_int(0)
        self.check(self.lib.sl_lyapunov_region(self.handle, _ptr(d_values), int(start), _ptr(d_work),
                                               _ptr(d_region), C.byref(sweeps)), "sl_lyapunov_region")
        return sweeps.value

    def bits_to_bytes(self, n, d_bits, d_bytes):
        self.check(self.lib.sl_bits_to_bytes(self.handle, n, _ptr(d_bits), _ptr(d_bytes)),
                   "sl_bits_to_bytes")

    def bits_count(self, n, d_bits, d_block_counts, d_offsets):
        """-> number of set bits among the first ``n`` (``sl_bits_count``; fills the two scratch arrays)."""
        total = C.c_int64(0)
        self.check(self.lib.sl_bits_count(self.handle, n, _ptr(d_bits), _ptr(d_block_counts), _ptr(d_offsets),
                                          C.byref(total)), "sl_bits_count")
        return total.value

    def bits_to_indices(self, n, d_bits, d_offsets, d_indices):
        if d_indices.numel() == 0:               # no bit set: nothing to write
            return
        self.check(self.lib.sl_bits_to_indices(self.handle, n, _ptr(d_bits), _ptr(d_offsets), _ptr(d_indices)),
                   "sl_bits_to_indices")

    def bytes_to_bits(self, n, d_bytes, d_bits):
        self.check(self.lib.sl_bytes_to_bits(self.handle, n, _ptr(d_bytes), _ptr(d_bits)),
                   "sl_bytes_to_bits")

    def bellman_sweep(self, lo, hi, actions, d_v_new, d_argmax, d_q, d_stats):
        if actions is None:
            n_act, pa = 0, None
        else:
            actions, pa = _as_c(actions)
            n_act = actions.shape[0]
        self.check(self.lib.sl_bellman_sweep(self.handle, lo, hi, n_act, pa, _ptr(d_v_new),
                                             _ptr(d_argmax), _ptr(d_q), _ptr(d_stats)),
                   "sl_bellman_sweep")

    def successor_cache_configure(self, max_bytes):
        """Budget of the Bellman sweeps' successor cache (``sl_successor_cache_configure``):
        negative = default (a quarter of the device's memory), 0 = no cache."""
        self.check(self.lib.sl_successor_cache_configure(self.handle, int(max_bytes)),
                   "sl_successor_cache_configure")

    def successor_cache_info(self):
        """``sl_successor_cache_info`` as a dict."""
        stats = SuccessorCacheStats()
        self.check(self.lib.sl_successor_cache_info(self.handle, C.byref(stats)),
                   "sl_successor_cache_info")
        return {name: int(getattr(stats, name)) for name, _ in SuccessorCacheStats._fields_}

    def eval_points(self, what, n, d_points, d_out):
        self.check(self.lib.sl_eval_points(self.handle, what, n, _ptr(d_points), _ptr(d_out)),
                   "sl_eval_points")

    def synchronize(self):
        self.check(self.lib.sl_ctx_synchronize(self.handle), "sl_ctx_synchronize")

    TIMING_LYAP_SWEEP, TIMING_FINALIZE, TIMING_BELLMAN = 0, 1, 2

    def timing_configure(self, slots):
        """``sl_timing_configure``: the library records HIP events around up to ``slots`` calls of
        each of its sweep entry points (0: off)."""
        self.check(self.lib.sl_timing_configure(self.handle, int(slots)), "sl_timing_configure")
        self._timing_slots = int(slots)

    def timing_collect(self, channel):
        """Durations (ms) of the calls recorded on ``channel`` since the last collect."""
        slots = getattr(self, "_timing_slots", 0)
        if not slots:
            return []
        out = (C.c_double * slots)()
        count = C.c_int(0)
        self.check(self.lib.sl_timing_collect(self.handle, int(channel), out, slots, C.byref(count)),
                   "sl_timing_collect")
        return [float(out[i]) for i in range(count.value)]

    def last_kernel(self):
        """Name of the kernel(s) the last sweep of this context launched (``sl_last_kernel``)."""
        name = self.lib.sl_last_kernel(self.handle)
        return name.decode() if name else ""

    # ---- RCCL collectives of the C ABI (the package itself uses torch.distributed) ---------
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.sl_comm_unique_id(buf)
        if rc != 0:
            raise HipEngineError("sl_comm_unique_id failed: %s" % lib.sl_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self.check(self.lib.sl_comm_init(self.handle, unique_id, rank, world), "sl_comm_init")

    def comm_destroy(self):
        self.check(self.lib.sl_comm_destroy(self.handle), "sl_comm_destroy")

    def allreduce_result(self, d_result):
        self.check(self.lib.sl_allreduce_result(self.handle, _ptr(d_result)), "sl_allreduce_result")

    def allgather(self, d_send, d_recv, nbytes):
        self.check(self.lib.sl_allgather(self.handle, _ptr(d_send), _ptr(d_recv), nbytes),
                   "sl_allgather")

    def allreduce_sum_u64(self, d_values, count):
        self.check(self.lib.sl_allreduce_sum_u64(self.handle, _ptr(d_values), count),
                   "sl_allreduce_sum_u64")

    def allreduce_max_f64(self, d_values, count):
        self.check(self.lib.sl_allreduce_max_f64(self.handle, _ptr(d_values), count),
                   "sl_allreduce_max_f64")

    # ---- diagnostics ---------------------------------------------------------------------
    def debug_mfma(self, a, b):
        a, pa = _as_c(a)
        b, pb = _as_c(b)
        out = np.zeros((16, 16))
        self.check(self.lib.sl_debug_mfma(self.handle, pa, pb, out.ctypes.data_as(c_double_p)),
                   "sl_debug_mfma")
        return out

    def debug_mfma4(self, a, b, c, mode=0):
        """v_mfma_f64_4x4x4_4b_f64 on per-lane operands of shape (nwaves, 64)."""
        a, pa = _as_c(np.atleast_2d(a))
        b, pb = _as_c(np.atleast_2d(b))
        c, pc = _as_c(np.atleast_2d(c))
        out = np.zeros_like(a)
        self.check(self.lib.sl_debug_mfma4(self.handle, a.shape[0], pa, pb, pc, mode,
                                           out.ctypes.data_as(c_double_p)), "sl_debug_mfma4")
        return out

    def debug_fp64_rate(self, which, iters=20000):
        out = (C.c_double * 3)()
        self.check(self.lib.sl_debug_fp64_rate(self.handle, which, iters, out),
                   "sl_debug_fp64_rate")
        return {"tflops": out[0], "shader_mhz": out[1], "cycles_per_slot": out[2]}


_default_context = None


def default_context():
    """Process-wide context on torch's current device."""
    global _default_context
    if _default_context is None:
        _default_context = Context()
    return _default_context
