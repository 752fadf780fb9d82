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
POLICY_LINEAR, POLICY_CONST, POLICY_TABLE, POLICY_TRI = 1, 2, 3, 4
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


class ModelDesc(C.Structure):
    _fields_ = [("grid", GridDesc), ("policy", PolicyDesc), ("dynamics", DynamicsDesc),
                ("value", ValueDesc), ("lipschitz", LipschitzDesc), ("reward", ValueDesc),
                ("gamma", C.c_double)]


class Key(C.Structure):
    _fields_ = [("vbits", C.c_uint64), ("index", C.c_int64)]


# sl_sweep_result as int64 words (a torch int64[8] tensor backs it on the device)
RESULT_WORDS = 8
R_FAIL_V, R_FAIL_I, R_LAST_V, R_LAST_I, R_MAX_V, R_MAX_I, R_BELOW, R_SAFE = range(8)

EXPORTS = [
    "sl_version", "sl_ctx_create", "sl_ctx_destroy", "sl_last_error", "sl_ctx_synchronize",
    "sl_last_kernel",
    "sl_model_set", "sl_gp_set_head", "sl_gp_append_point", "sl_gp_configure", "sl_tri_set", "sl_tri_set_table",
    "sl_network_set", "sl_values", "sl_lyap_sweep", "sl_lyap_finalize", "sl_select_pass",
    "sl_bits_to_bytes", "sl_bytes_to_bits", "sl_bellman_sweep", "sl_eval_points",
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
    lib.sl_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.sl_ctx_destroy.argtypes = [C.c_void_p]
    lib.sl_ctx_synchronize.argtypes = [C.c_void_p]
    lib.sl_model_set.argtypes = [C.c_void_p, C.POINTER(ModelDesc)]
    lib.sl_gp_set_head.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   c_double_p, c_double_p, c_double_p, C.c_double, c_double_p]
    lib.sl_gp_append_point.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]
    lib.sl_gp_configure.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.sl_tri_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(GridDesc), C.c_int,
                               C.POINTER(C.c_int32), c_double_p, c_double_p, C.c_int, C.c_int,
                               C.c_void_p]
    lib.sl_tri_set_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.sl_network_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), c_double_p]
    lib.sl_values.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.sl_lyap_sweep.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_lyap_finalize.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, Key, Key, C.c_void_p, C.c_void_p]
    lib.sl_select_pass.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                   C.c_uint64, C.c_uint64, C.c_void_p]
    lib.sl_bits_to_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sl_bytes_to_bits.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sl_bellman_sweep.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, c_double_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sl_eval_points.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
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
        if name not in ("sl_last_error", "sl_last_kernel"):
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

    def bits_to_bytes(self, n, d_bits, d_bytes):
        self.check(self.lib.sl_bits_to_bytes(self.handle, n, _ptr(d_bits), _ptr(d_bytes)),
                   "sl_bits_to_bytes")

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

    def eval_points(self, what, n, d_points, d_out):
        self.check(self.lib.sl_eval_points(self.handle, what, n, _ptr(d_points), _ptr(d_out)),
                   "sl_eval_points")

    def synchronize(self):
        self.check(self.lib.sl_ctx_synchronize(self.handle), "sl_ctx_synchronize")

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
