"""``Lyapunov``: safe-set computation on a GridWorld, executed by the HIP engine.

Drop-in for the constructor / ``update_values`` / ``update_safe_set`` surface of
``safe_learning/lyapunov.py:142-606``.  What changes underneath:

* the reference sorts all cells by V (``np.argsort``, ``lyapunov.py:512``) and walks them in
  10 000-cell batches through a TensorFlow graph until the first failing cell
  (``:524-587``).  Here one fused kernel checks EVERY cell of this rank's grid shard and
  reduces the lexicographically smallest failing ``(V, index)`` key; a second streaming pass
  marks ``safe_i = init_i or key_i < key*``.  Both formulations give the same mask (see
  DESIGN.md, "prefix rule"); ties in V are ordered by flat index (the oracle's definition).
* ``c_max`` follows the reference's index arithmetic, including its two quirks
  (``lyapunov.py:590-595``): "first cell fails" reads the largest value, "nothing fails" reads
  the value just before the last batch - obtained with a device-side radix select.
* with ``torch.distributed`` initialised, every rank owns a contiguous 64-aligned index range
  and the reductions become RCCL collectives of a few bytes.
"""

import numpy as np

from . import _hip
from . import distributed as dist_utils
from ._model import ModelBuilder
from .configuration import config

__all__ = ['Lyapunov', 'smallest_boundary_value', 'get_safe_sample', 'perturb_actions',
           'get_lyapunov_region']

def _digest(array):
    """128-bit content digest of an array's bytes (xxh3 at several GB/s when the xxhash package is
    there, blake2b otherwise): how in-place edits of ``initial_safe_set`` and ``safe_set`` are
    noticed - the reference reads both afresh on every call (``lyapunov.py:500-510``)."""
    data = np.ascontiguousarray(array)
    view = memoryview(data.reshape(-1).view(np.uint8)) if data.size else b''
    try:
        import xxhash
        return xxhash.xxh3_128_intdigest(view)
    except ImportError:                                  # pragma: no cover - image has xxhash
        import hashlib
        return hashlib.blake2b(view, digest_size=16).digest()


def _frozen(array):
    """True for an ndarray nobody can edit in place through NumPy: read-only, and so is every array
    it is a view of.  Such a mask is identified by object identity; a writable one by its digest."""
    while isinstance(array, np.ndarray):
        if array.flags.writeable:
            return False
        array = array.base
    return array is None


_U64_MAX = (1 << 64) - 1
_I64_MAX = (1 << 63) - 1
_KEY_NONE = (_U64_MAX, _I64_MAX)


def vbits_to_float(vbits):
    """Inverse of the kernels' order-preserving float64 -> uint64 map (``sl_vbits``)."""
    if vbits == _U64_MAX:
        return float('nan')
    raw = (vbits & ((1 << 63) - 1)) if vbits & (1 << 63) else (~vbits) & _U64_MAX
    return float(np.array([raw], dtype=np.uint64).view(np.float64)[0])


def smallest_boundary_value(fun, discretization):
    """Smallest value of the quadratic / table function ``fun`` on the grid boundary
    (``lyapunov.py:22-56``), evaluated with the engine's value pass."""
    lyap = Lyapunov.__new__(Lyapunov)
    lyap._bare_init(discretization, fun)
    values = lyap.values.reshape(discretization.num_points)
    best = np.inf
    for axis in range(discretization.ndim):
        face = np.take(values, [0, -1], axis=axis)
        best = min(best, float(face.min()))
    return best


class Lyapunov(object):
    """See ``safe_learning/lyapunov.py:142-175`` for the argument meanings.

    ``lyapunov_function``, ``dynamics``, ``policy`` and ``lipschitz_lyapunov`` are specs from
    :mod:`safe_learning_amd.functions` (arbitrary Python callables cannot run in a kernel).
    """

    def __init__(self, discretization, lyapunov_function, dynamics, lipschitz_dynamics,
                 lipschitz_lyapunov, tau, policy, initial_set=None, adaptive=False):
        self._setup_engine(discretization)
        self.policy = policy
        self.tau = tau
        self.dynamics = dynamics
        self.lyapunov_function = lyapunov_function
        self._lipschitz_dynamics = lipschitz_dynamics
        self._lipschitz_lyapunov = lipschitz_lyapunov
        self.adaptive = adaptive
        self._refinement_host = None
        self._pending = None
        self._safe_count = 0
        self.c_max = 0.
        self.feed_dict = _CMaxView(self)
        self.initial_safe_set = initial_set
        # safe_set starts as the initial set (lyapunov.py:187-192)
        self._safe_host = np.zeros(self.discretization.nindex, dtype=bool)
        if initial_set is not None:
            self._safe_host[initial_set] = True
        self._safe_host_valid = True
        self._safe_dev_valid = False
        self._safe_host_digest = None
        self.update_values()

    # ---- engine plumbing -----------------------------------------------------------------
    def _setup_engine(self, discretization):
        import torch
        self.discretization = discretization
        self._ctx = _hip.Context()
        self._builder = ModelBuilder(self._ctx, discretization)
        n = discretization.nindex
        self._rank, self._world = dist_utils.rank_and_world()
        # collectives run with more than one rank (or SL_FORCE_COLLECTIVES=1: the same call
        # sequence at world size 1, to exercise RCCL on a one-GPU box)
        self._collective = dist_utils.is_distributed()
        self._bounds = dist_utils.shard_bounds(n, self._world)
        self._lo, self._hi = self._bounds[self._rank], self._bounds[self._rank + 1]
        dev = self._ctx.torch_device
        count = self._hi - self._lo
        self._nwords = (count + 63) // 64
        # every rank's buffers have the capacity of a full shard (``per`` cells, a multiple of 64),
        # so that the gathers are ONE all_gather_into_tensor of equal pieces without a pad copy
        per = max(self._bounds[1] - self._bounds[0], 1)
        cap, wcap = (per, -(-per // 64)) if self._collective else (max(count, 1), max(self._nwords, 1))
        self._values_capacity = cap
        self._d_values_buffer = None    # V of this rank's shard: allocated when something reads it
        self._values_stale = True
        self._values_implicit = False   # quadratic V: the passes recompute it from the cell index
        self._d_init = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_neg = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_safe = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_result = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
        self._d_folded = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
        self._d_select = torch.zeros(_hip.SELECT_WORDS, dtype=torch.int64, device=dev)
        self._d_hist = torch.zeros(256, dtype=torch.int64, device=dev)
        self._values_host = None
        self._d_values_full = None      # all shards of V: gathered on demand (gather_values)
        self._d_safe_full = None        # all shards' mask words, gathered by update_safe_set
        self._init_version = None
        self._init_object = None
        self._safe_host_digest = None   # digest of the host mask when it last equalled the device's
        self.mask_uploads = 0           # host -> device copies of grid-sized masks (tests, bench)

    def _bare_init(self, discretization, fun):
        """Value-only instance used by ``smallest_boundary_value``."""
        from .functions import ConstantFunction, LinearSystem
        self._setup_engine(discretization)
        d = discretization.ndim
        self.lyapunov_function = fun
        self.policy = ConstantFunction(np.zeros(1))
        self.dynamics = LinearSystem((np.eye(d), np.zeros((d, 1))))
        self._lipschitz_dynamics = 0.
        self._lipschitz_lyapunov = 0.
        self.tau = 0.
        self.update_values()

    def _upload_model(self):
        self._builder.upload(self.policy, self.dynamics, self.lyapunov_function,
                             self._lipschitz_lyapunov, self._lipschitz_dynamics, self.tau)

    @property
    def _d_values(self):
        """V on this rank's shard as a device tensor (``sl_values``), computed when first read:
        with a quadratic V the sweep and the streaming passes recompute the ordering keys from the
        cell index (``sl_values_implicit``) and the 8 bytes per cell are neither written nor read."""
        import torch
        if self._d_values_buffer is None:
            self._d_values_buffer = torch.zeros(self._values_capacity, dtype=torch.float64,
                                                device=self._ctx.torch_device)
            self._values_stale = True
        if self._values_stale:
            self._values_into(self._lo, self._hi, self._d_values_buffer)
            self._values_stale = False
        return self._d_values_buffer

    def _values_into(self, lo, hi, buffer):
        """V of the cells ``[lo, hi)`` (``sl_values``) into a device buffer."""
        # implicit values are those of the function as it was at update_values() (a copy of
        # its d x d matrix): the reference's array does not follow later edits of the function
        snapshot = getattr(self, '_implicit_snapshot', None)
        fun = snapshot[0] if (self._values_implicit and snapshot is not None) else self.lyapunov_function
        self._builder.upload(self.policy, self.dynamics, fun, self._lipschitz_lyapunov,
                             self._lipschitz_dynamics, self.tau)
        self._ctx.values(lo, hi, buffer)
        if fun is not self.lyapunov_function:
            self._upload_model()

    def _implicit_signature(self):
        """What the kernels' recomputed keys depend on (closed-form V: its matrix and sign)."""
        fun = self._lyapunov_function
        matrix = getattr(fun, 'matrix', None)
        return (type(fun).__name__, bool(getattr(fun, 'negate', False)),
                None if matrix is None else np.asarray(matrix).tobytes())

    def _values_arg(self):
        """What the passes get as ``d_values``: None when the keys are recomputed in the kernels.

        ``values`` are the numbers of the last ``update_values()`` (``lyapunov.py:305-322``): if the
        closed-form function was edited IN PLACE since then (same object, other matrix), the keys
        the kernels would recompute are no longer those numbers - V is written down from the
        snapshot taken at ``update_values()`` and the passes read the array."""
        snapshot = getattr(self, '_implicit_snapshot', None)
        if self._values_implicit and snapshot is not None and self._implicit_signature() != snapshot[1]:
            self._d_values
            self._values_implicit = False
        return None if self._values_implicit else self._d_values

    # ---- reference attribute surface -----------------------------------------------------
    @property
    def lyapunov_function(self):
        return self._lyapunov_function

    @lyapunov_function.setter
    def lyapunov_function(self, fun):
        # In the reference ``values`` keeps the OLD function's numbers until update_values() is
        # called (lyapunov.py:305-322).  If the kernels currently recompute the keys from the
        # model, write them down with the old function before the model changes.
        old = getattr(self, '_lyapunov_function', None)
        if old is not None and fun is not old and getattr(self, '_values_implicit', False):
            self._d_values                   # (from the snapshot of update_values(), see there)
            self._values_implicit = False
        self._lyapunov_function = fun

    @property
    def c_max(self):
        """Level of the safe set, ``values[order[max_index]]`` (``lyapunov.py:590-595``).

        One process, ``can_shrink=True``: ``update_safe_set`` returns as soon as its kernels are
        enqueued - the mask is complete on the device, and the one number the HOST needs from the
        64-byte result record is fetched here, on the first read (a sweep of C2's 65 536 cells is
        0.3 ms: a synchronous copy and the host code behind it were a fifth of the step)."""
        self._resolve_pending()
        return self._c_max

    @c_max.setter
    def c_max(self, value):
        self._pending = None
        self._c_max = value

    @property
    def safe_count(self):
        """Cells of the safe set (all ranks) without a copy of the mask."""
        self._resolve_pending()
        return self._safe_count

    @safe_count.setter
    def safe_count(self, value):
        self._safe_count = value

    def _resolve_pending(self):
        pending, self._pending = getattr(self, '_pending', None), None
        if pending is None:
            return
        folded, n, batch = pending
        stats = {}
        self._upload_model()           # (the no-failure quirk runs select passes over the keys)
        self._c_max = prefix_rule_finish(_HipShardEngine(self), folded, n, batch, True, stats)
        self._safe_count = stats['safe']

    @property
    def initial_safe_set(self):
        return self._initial_safe_set

    @initial_safe_set.setter
    def initial_safe_set(self, value):
        self._initial_safe_set = value
        self._init_version = None
        self._init_object = None

    def lipschitz_dynamics(self, states):
        """Lipschitz constant of the dynamics at ``states`` (``lyapunov.py:227-244``): the scalar,
        or ``[n, 1]`` for the state-dependent spec ``c + Norm1Function(LinearSystem(M))``
        (host arithmetic in the kernels' operation order)."""
        lf = self._lipschitz_dynamics
        if np.isscalar(lf):
            return lf
        states = np.atleast_2d(np.asarray(states, dtype=np.float64))
        rows = lf.fun.matrix
        acc = None
        for r in range(rows.shape[0]):
            t = states[:, 0] * rows[r, 0]
            for k in range(1, states.shape[1]):
                t = t + states[:, k] * rows[r, k]
            acc = np.abs(t) if acc is None else acc + np.abs(t)
        return (lf.constant + acc)[:, None]

    def lipschitz_lyapunov(self, states):
        """``L_v`` at explicit states (``lyapunov.py:246-263``): the scalar, or ``[n, cols]``."""
        if np.isscalar(self._lipschitz_lyapunov):
            return self._lipschitz_lyapunov
        from . import _evaluate
        return _evaluate.value(self.lyapunov_function, states, self._lipschitz_lyapunov)[1]

    def threshold(self, states, tau=None):
        """``-|L_v(states)|_1 (1 + L_f) tau`` (``lyapunov.py:265-288``), computed on the host from
        the device-evaluated ``L_v`` (same operation order as the kernels)."""
        tau = self.tau if tau is None else tau
        lv = self.lipschitz_lyapunov(states)
        if not np.isscalar(lv) and lv.shape[1] > 1:
            acc = np.abs(lv[:, 0])
            for k in range(1, lv.shape[1]):
                acc = acc + np.abs(lv[:, k])
            lv = acc[:, None]
        return (-lv) * (1. + self.lipschitz_dynamics(states)) * tau

    def v_decrease_confidence(self, states, next_states):
        """``(V(next) - V(states), sum_j L_v(next)_j error_j)`` (``lyapunov.py:324-354``)."""
        from . import _evaluate
        if isinstance(next_states, (tuple, list)):
            next_states, error_bounds = next_states
            lv = self.lipschitz_lyapunov(next_states)
            prod = lv * np.asarray(error_bounds)
            bound = prod[:, [0]].copy()
            for k in range(1, prod.shape[1]):
                bound = bound + prod[:, [k]]
        else:
            bound = 0.
        v_decrease = (_evaluate.value(self.lyapunov_function, next_states)
                      - _evaluate.value(self.lyapunov_function, states))
        return v_decrease, bound

    def v_decrease_bound(self, states, next_states):
        """``lyapunov.py:356-376``."""
        v_dot, v_dot_error = self.v_decrease_confidence(states, next_states)
        return v_dot + v_dot_error

    @property
    def values(self):
        """V at every grid point, ``float64[nindex]`` (``lyapunov.py:305-322``).

        With more than one rank every rank keeps only its shard of V on the device (the sweep
        needs nothing else).  Reading the attribute is a LOCAL operation on every rank: each rank
        holds the whole model, so the first read after ``update_values`` computes V of the whole
        grid on this rank's GPU (``sl_values``, 0.75 ms at 128^4) - no collective hides behind the
        attribute, a rank-guarded read cannot deadlock (round 5 gathered the shards here)."""
        if self._values_host is None:
            self._values_host = self.gather_values().cpu().numpy()
        return self._values_host

    def gather_values(self):
        """V of the whole grid as one device tensor ``float64[nindex]`` (cached until the next
        ``update_values``).  On a sharded grid the other ranks' cells are COMPUTED here (every
        rank has the model; 2.1 GB per GPU at 128^4, which is why it is not done eagerly) - a
        local operation, not a collective."""
        if not self._collective:
            return self._d_values[:self._hi - self._lo]
        if self._d_values_full is None:
            import torch
            n = self.discretization.nindex
            full = torch.empty(n, dtype=torch.float64, device=self._ctx.torch_device)
            self._values_into(0, n, full)
            self._d_values_full = full
        return self._d_values_full

    def _safe_full_buffer(self):
        """Pre-sized receive buffer of the mask gather (world x words per shard), reused."""
        import torch
        need = self._world * self._d_safe.numel()
        buf = getattr(self, '_d_safe_gather', None)
        if buf is None or buf.numel() != need:
            buf = self._d_safe_gather = torch.empty(need, dtype=torch.int64,
                                                    device=self._ctx.torch_device)
        return buf

    @property
    def safe_set(self):
        """``bool[nindex]`` mask, the same array object across calls (``lyapunov.py:187, 598-606``).
        A local read on every rank (``update_safe_set`` gathers the shards' mask words).

        Reading does not give up the device copy: the array's digest is kept, and the next call
        that needs the mask on the device (``update_safe_set(can_shrink=False)``,
        ``get_safe_sample``) compares it - an array the caller edited in place is packed and
        uploaded again, an untouched one (the notebooks' loop looks at the set after every update)
        costs one hash of the host array instead of a 268 MB round trip at 128^4."""
        if not self._safe_host_valid:
            import torch
            n = self.discretization.nindex
            words = self._d_safe_full if self._d_safe_full is not None else self._d_safe
            d_bytes = torch.empty(max(-(-n // 8) * 8, 8), dtype=torch.uint8,
                                  device=self._ctx.torch_device)
            self._ctx.bits_to_bytes(n, words, d_bytes)
            self._safe_host[:] = d_bytes[:n].cpu().numpy().astype(bool)
            self._safe_host_valid = True
            # the caller may edit the array from now on: remember what the device copy equals
            self._safe_host_digest = _digest(self._safe_host)
        return self._safe_host

    @safe_set.setter
    def safe_set(self, value):
        self._safe_host[:] = value
        self._safe_host_valid = True
        self._safe_dev_valid = False
        self._safe_host_digest = None

    def _safe_device_current(self):
        """Is the device copy of the safe set still the truth?  (False: the host array is - it was
        assigned, or edited in place since it was downloaded.)"""
        if self._safe_dev_valid and self._safe_host_valid:
            if self._safe_host_digest is None or _digest(self._safe_host) != self._safe_host_digest:
                self._safe_dev_valid = False
                self._safe_host_digest = None
        return self._safe_dev_valid

    def _sync_safe_to_device(self):
        """This rank's words of the previous safe set on the device (``lyapunov.py:507-510``)."""
        if not self._safe_device_current():
            self._upload_mask(self._safe_host, self._d_safe)
            self._d_safe_full = None            # (collective runs: gathered again by the next update)
            self._safe_dev_valid = True
            self._safe_host_digest = _digest(self._safe_host)

    @property
    def _refinement(self):
        """Refinement N(x) per cell (``lyapunov.py:220-225``): the array kept by the adaptive
        branch, otherwise 1 on safe cells and 0 elsewhere (``:531, 586, 601-606``)."""
        if self._refinement_host is None and getattr(self, '_refinement_full', None) is not None:
            # (the shards were gathered INSIDE update_safe_set, which every rank calls: reading the
            # attribute is local - no collective behind a property)
            self._refinement_host = self._refinement_full.cpu().numpy().astype(int)
        if self._refinement_host is not None:
            return self._refinement_host
        return self.safe_set.astype(int)

    def is_safe(self, state):
        """``lyapunov.py:290-303``."""
        return self.safe_set[self.discretization.state_to_index(state)]

    def decrease_bits(self, policy, include_initial=True):
        """Device bit words (int64, this rank's shard) of the cells where the decrease condition
        holds under ``policy`` - a per-vertex action array ``[nindex, m]`` or a policy spec - OR-ed
        with the initial safe set if asked for.  One sweep of the engine's kernel; the object's own
        policy, safe set and ``c_max`` are left alone."""
        import torch
        self._resolve_pending()              # (the sweep below reuses the result record)
        own_policy = self.policy
        self.policy = policy
        try:
            self._upload_model()
            self._refresh_init_bits()
            d_neg = torch.zeros_like(self._d_neg)
            self._ctx.lyap_sweep(self._lo, self._hi, self._d_init, self._values_arg(), d_neg,
                                 self._d_result)
        finally:
            self.policy = own_policy
            self._upload_model()
        if include_initial and self._initial_safe_set is not None:
            d_neg |= self._d_init
        return d_neg

    def safety_constraint(self, policy, include_initial=True):
        """``bool[nindex]``: where the decrease condition holds under ``policy`` - a per-vertex
        action array ``[nindex, m]`` or a policy spec (``lyapunov.py:378-406``).  The reference
        method cannot run as written (it compares with the bound method ``self.threshold`` and
        hands the discretization object to the dynamics); this is what its docstring describes,
        evaluated by the same sweep kernel as ``update_safe_set``.

        ``PolicyIteration.discrete_policy_optimization(actions, constraint=lyapunov)`` uses the
        device form (:meth:`decrease_bits`, one sweep per action) and never builds this array."""
        import torch
        d_neg = self.decrease_bits(policy, include_initial)
        count = self._hi - self._lo
        d_bytes = torch.empty(max(-(-count // 8) * 8, 8), dtype=torch.uint8,
                              device=self._ctx.torch_device)
        self._ctx.bits_to_bytes(count, d_neg, d_bytes)
        sizes = [self._bounds[r + 1] - self._bounds[r] for r in range(self._world)]
        return dist_utils.allgather_concat(d_bytes[:count], sizes).cpu().numpy().astype(bool)

    # ---- device helpers ------------------------------------------------------------------
    def _upload_mask(self, host_mask, d_bits):
        """bool[nindex] host mask -> this rank's bit words."""
        import torch
        count = self._hi - self._lo
        if count == 0:
            return
        chunk = np.ascontiguousarray(host_mask[self._lo:self._hi]).view(np.uint8)
        self.mask_uploads += 1
        import warnings
        with warnings.catch_warnings():          # (a read-only mask is only read here)
            warnings.simplefilter("ignore", UserWarning)
            d_bytes = torch.from_numpy(chunk).to(self._ctx.torch_device)
        self._ctx.bytes_to_bits(count, d_bytes, d_bits)

    def _refresh_init_bits(self):
        """Bit words of the initial safe set on the device, uploaded again whenever the set has
        changed - ALSO when it was edited in place: the reference reads ``initial_safe_set`` afresh
        on every call (``lyapunov.py:500-506, 604-606``).  A writable array (or a list) is
        identified by a digest of its content, computed on every call (xxh3: 0.15 ms per MB); an
        array that nobody can write to (``mask.flags.writeable = False``, no writable base) by
        object identity - the way to keep a 268 MB mask at 128^4 from being hashed by every
        update.  Caveat of the identity rule: NumPy lets the owner flip ``flags.writeable`` back on,
        and a writable VIEW taken before the freeze stays writable - edits made that way after the
        first use go unnoticed (assign a new array, or keep the mask writable, if it must change)."""
        init = self._initial_safe_set
        if init is None:
            version = ('none',)
        elif _frozen(init):
            # read-only: identified by object identity from its FIRST use on - never hashed, no warning
            if init is self._init_object:
                return
            arr = np.asarray(init)
            version = ('frozen',)
        else:
            arr = np.asarray(init)
            if arr.nbytes > (1 << 24) and not getattr(self, '_warned_large_mask', False):
                import warnings
                self._warned_large_mask = True
                warnings.warn("initial_safe_set is a writable array of %d MB: its content is hashed on every "
                              "update_safe_set (about 0.15 ms per MB) so that in-place edits are noticed like "
                              "in the reference; make it read-only (mask.flags.writeable = False) to have it "
                              "identified by object identity instead" % (arr.nbytes >> 20), RuntimeWarning,
                              stacklevel=3)
            version = (arr.shape, arr.dtype.str, _digest(arr))
            # the cached object is held strongly and compared by identity: an id() could be recycled
            if version == self._init_version and init is self._init_object:
                return
        if init is None:
            if self._init_version == version:
                return
            self._d_init.zero_()
        else:
            if arr.dtype == bool and arr.shape == (self.discretization.nindex,):
                mask = arr
            else:
                mask = np.zeros(self.discretization.nindex, dtype=bool)
                mask[init] = True
            self._upload_mask(mask, self._d_init)
        self._init_version = version
        self._init_object = init

    # ---- reference methods ---------------------------------------------------------------
    def update_values(self, gather=False):
        """Recompute V on the grid (``lyapunov.py:305-322``).

        With a quadratic V nothing is computed here: the kernels recompute the ordering keys of
        ``lyapunov.py:512`` from the cell index and the array is materialised when ``values`` is
        read.  ``gather=True`` (sharded grids): gather all shards now, collectively, so that a
        later read of ``values`` on ONE rank only is a local read (the lazy gather behind the
        attribute is a collective and must otherwise be reached by every rank)."""
        import copy
        # c_max / safe_count of the last update_safe_set are the reference's values OF THAT MOMENT
        # (lyapunov.py:590-595): a deferred no-failure select reads the ordering keys, so it has to
        # run before they are replaced
        self._resolve_pending()
        self._upload_model()
        self._values_implicit = self._ctx.values_implicit()
        self._implicit_snapshot = ((copy.deepcopy(self._lyapunov_function), self._implicit_signature())
                                   if self._values_implicit else None)
        self._values_stale = True
        if not self._values_implicit:
            self._d_values                   # computes this rank's shard now
        self._values_host = None
        self._d_values_full = None       # shards only; ``values`` / ``gather_values`` gather lazily
        if gather:
            self.gather_values()

    def update_safe_set(self, can_shrink=True, max_refinement=1, safety_factor=1.,
                        parallel_iterations=1):
        """Recompute the safe set (``lyapunov.py:407-606``).  ``max_refinement > 1`` on an adaptive
        instance takes the adaptive branch (``:540-582``): the per-cell quantities still come from
        one GPU sweep, the sequential refinement bookkeeping runs on the host like the reference's."""
        if self.adaptive and max_refinement > 1:
            return self._update_safe_set_adaptive(can_shrink, max_refinement, safety_factor)
        # N(x) of an earlier ADAPTIVE update survives a non-adaptive one that must not shrink
        # (lyapunov.py:507-510): carried on the device below; otherwise N(x) is 1 on the safe set
        carry = None if can_shrink else self._refinement_shard()
        if carry is not None:
            carry = carry.clone()
        self._refinement_host = None
        self._refinement_dev = None
        self._refinement_full = None
        self._upload_model()
        self._refresh_init_bits()
        if not can_shrink:                                       # lyapunov.py:507-510
            self._sync_safe_to_device()
        engine = _HipShardEngine(self)
        stats, deferred = {}, ([] if (can_shrink and not self._collective) else None)
        self._pending = None                  # (an unread c_max of the previous update is history)
        n, batch = self.discretization.nindex, int(config.gp_batch_size)
        c_max = prefix_rule(engine, n, batch, can_shrink, stats, defer=deferred)
        if deferred:
            self._pending = (deferred[0], n, batch)    # c_max / safe_count: read on demand
        else:
            self.c_max = c_max
            self.safe_count = stats['safe']   # cells in the safe set (all ranks), no mask copy
        if carry is not None:
            keep = stats.get('keep')
            self._ctx.refinement_carry(self._lo, self._hi, self._values_arg(), self._d_init, self._d_neg,
                                       stats['folded'],
                                       None if keep is None else keep[_hip.S_KEY_V:_hip.S_KEY_V + 2], carry)
            self._publish_refinement(carry)
        self._safe_host_valid = False
        self._safe_dev_valid = True
        self._safe_host_digest = None
        if self._collective:
            # shards start at multiples of 64 cells, so their mask words concatenate (4 MB per
            # rank at 128^4 over 8 GPUs); afterwards ``safe_set`` is a local read on every rank
            self._d_safe_full = dist_utils.allgather_equal(
                self._d_safe, -(-self.discretization.nindex // 64), out=self._safe_full_buffer())


    def _update_safe_set_adaptive(self, can_shrink, max_refinement, safety_factor):
        """Adaptive discretisation (``lyapunov.py:445-487, 540-582``), bug-compatible: as written
        in the reference the refined check compares the decrease of every cell of the candidate
        run with each cell's refined threshold ``threshold(x, tau / N(x))`` (the refined points
        themselves are never evaluated), see DESIGN.md.

        All of it runs in HIP kernels on sharded data (``adaptive_rule``): one sweep of this rank's
        cells, a device radix sort of the ``(V, index)`` keys instead of the host ``argsort``, every
        batch of the reference's loop judged in parallel, rows exchanged between the owner of a cell
        and the owner of its sorted position with two all-to-alls; no rank holds anything of grid
        size but its own shard."""
        self._resolve_pending()
        self._upload_model()
        self._refresh_init_bits()
        if not can_shrink:                                       # lyapunov.py:507-510
            self._sync_safe_to_device()
        engine = _HipAdaptiveEngine(self, can_shrink)
        stats = {}
        self.c_max = adaptive_rule(engine, self.discretization.nindex, int(config.gp_batch_size),
                                   max_refinement, max(float(safety_factor), 1.), stats)
        self.safe_count = stats['safe']
        self._publish_refinement(engine.refinement)     # this rank's shard, int64
        self._safe_host_valid = False
        self._safe_dev_valid = True
        self._safe_host_digest = None
        if self._collective:
            self._d_safe_full = dist_utils.allgather_equal(
                self._d_safe, -(-self.discretization.nindex // 64), out=self._safe_full_buffer())

    def _publish_refinement(self, shard):
        """This rank's refinement shard (device, or None) becomes the object's; with more than one
        rank the shards are all-gathered HERE, inside the update every rank takes part in."""
        self._refinement_dev = shard
        self._refinement_host = None
        self._refinement_full = None
        if shard is not None:
            sizes = [self._bounds[r + 1] - self._bounds[r] for r in range(self._world)]
            self._refinement_full = dist_utils.allgather_concat(shard, sizes)

    def _refinement_shard(self):
        """Refinement N(x) of this rank's cells as a device tensor (int64), or None when it is
        simply 1 on safe cells (no adaptive update yet)."""
        import torch
        dev = self._ctx.torch_device
        if getattr(self, '_refinement_dev', None) is not None:
            return self._refinement_dev
        if self._refinement_host is not None:
            return torch.from_numpy(np.ascontiguousarray(
                np.asarray(self._refinement_host, dtype=np.int64)[self._lo:self._hi])).to(dev)
        return None


class _HipShardEngine(object):
    """This rank's shard of the grid as the prefix rule sees it (all work in HIP kernels, every
    intermediate result - records, folded records, the radix-select state - in device memory)."""

    def __init__(self, lyap):
        self.lyap = lyap
        self.prior = None

    def sweep(self, can_shrink):
        """Decrease check of every cell; the record's ``fail`` is the local lexmin failing
        ``(vbits, index)``."""
        ly = self.lyap
        self.prior = ly._d_init if can_shrink else ly._d_safe.clone()     # lyapunov.py:500-510
        ly._ctx.lyap_sweep(ly._lo, ly._hi, self.prior, ly._values_arg(), ly._d_neg, ly._d_result)
        return ly._d_result

    def fold(self, records, count):
        """``count`` gathered records (device) -> one: lexmin / lexmax of the keys, sums of the
        counters (``sl_fold_results``)."""
        ly = self.lyap
        if count == 1:              # one shard: the fold of one record is that record (no launch)
            return records
        ly._ctx.fold_results(records, count, ly._d_folded)
        return ly._d_folded

    def finalize(self, folded, keep_state, use_prior):
        """``safe = init | key < key* | (prior & key >= key_keep)`` with ``key*`` = the folded
        record's ``fail`` and ``key_keep`` = the select state's key, both read by the kernel."""
        ly = self.lyap
        keep = None if keep_state is None else keep_state[_hip.S_KEY_V:_hip.S_KEY_V + 2]
        # (the record the pass reads and the one it writes are two buffers)
        out = ly._d_folded if folded.data_ptr() == ly._d_result.data_ptr() else ly._d_result
        ly._ctx.lyap_finalize_dev(ly._lo, ly._hi, ly._values_arg(), ly._d_init,
                                  self.prior if use_prior else None, folded, keep, ly._d_safe, out)
        return out

    def _many(self, slot):
        """State / histogram rows of select number ``slot`` of a multi-k select (one block each, so
        that the histograms of all k are ONE tensor for the all-reduce)."""
        import torch
        ly = self.lyap
        have = getattr(self, '_many_states', None)
        if have is None or have.shape[0] <= slot:
            rows = max(8, 2 * (slot + 1))
            states = torch.zeros((rows, _hip.SELECT_WORDS), dtype=torch.int64, device=ly._ctx.torch_device)
            hists = torch.zeros((rows, 256), dtype=torch.int64, device=ly._ctx.torch_device)
            if have is not None:
                states[:have.shape[0]] = have
            self._many_states, self._many_hists = states, hists
        return self._many_states[slot], self._many_hists[slot]

    def select_begin(self, k, batch, folded, n, slot=None):
        ly = self.lyap
        state = ly._d_select if slot is None else self._many(slot)[0]
        ly._ctx.select_begin(state, k, batch, folded, n)
        return state

    def select_hist(self, which, byte, state, slot=None):
        """Local 256-bin histogram of one radix-select pass (int64[256], device)."""
        ly = self.lyap
        hist = ly._d_hist if slot is None else self._many(slot)[1]
        ly._ctx.select_hist(ly._lo, ly._hi, ly._values_arg(), which, byte, state, hist)
        return hist

    def stack_hists(self, hists):
        """The histograms of a multi-k pass as one tensor (they are rows of one block)."""
        return self._many_hists[:len(hists)]

    def select_digit(self, which, byte, hist, state):
        self.lyap._ctx.select_digit(which, byte, hist, state)


def select_kth(engine, k, batch, folded, n):
    """Select state (device) holding the k-th smallest ``(vbits, index)`` key (0-based) over all
    shards; ``k < 0``: ``k = (count_below // batch + 1) * batch`` taken from the folded record on
    the device.  Two 8-pass radix selects (value bytes, then index bytes among equal values), one
    2 KiB SUM all-reduce each, no host round trip: digits and prefixes stay on the device."""
    state = engine.select_begin(k, batch, folded, n)
    for which in (0, 1):
        for byte in range(7, -1, -1):
            hist = dist_utils.allreduce_sum_(engine.select_hist(which, byte, state))
            engine.select_digit(which, byte, hist, state)
    return state


def select_kth_many(engine, ks, batch, folded, n):
    """:func:`select_kth` for several ranks ``ks`` at once -> list of select states.  The sixteen
    passes of all selects run in lockstep and the histograms of one pass travel in ONE all-reduce
    (``len(ks)`` x 2 KiB): sixteen collectives for the ``world - 1`` splitters of the adaptive
    branch instead of sixteen per splitter (112 serialized all-reduces at world 8)."""
    states = [engine.select_begin(k, batch, folded, n, slot=r) for r, k in enumerate(ks)]
    if not states:
        return states
    for which in (0, 1):
        for byte in range(7, -1, -1):
            hists = [engine.select_hist(which, byte, state, slot=r) for r, state in enumerate(states)]
            block = dist_utils.allreduce_sum_(engine.stack_hists(hists))
            for r, state in enumerate(states):
                engine.select_digit(which, byte, block[r], state)
    return states


def _host_words(tensor):
    return [int(v) for v in tensor.detach().cpu().numpy().reshape(-1)]


def prefix_rule(engine, n, batch, can_shrink, stats=None, defer=None):
    """The safe-set rule of ``lyapunov.py:512-606`` over sharded cells; returns ``c_max``.

    ``engine`` owns one contiguous shard (the HIP engine in production, a NumPy stand-in in the
    CPU tests of the multi-rank path).  One sequence of launches and collectives: sweep ->
    all-gather of the 64-byte records -> fold (device) -> streaming pass that reads ``key*`` from
    the folded record -> all-gather -> fold -> ONE 64-byte copy to the host.  ``can_shrink=False``
    or the no-failure ``c_max`` quirk add a device-resident radix select (sixteen 2 KiB SUM
    all-reduces) and one more copy.

    ``defer`` (a list; one process, ``can_shrink=True``): the device part only - the folded record is
    appended and ``None`` returned; :func:`prefix_rule_finish` reads it when somebody asks for
    ``c_max`` (the safe set itself is complete on the device: nothing of it is decided on the host)."""
    folded = engine.fold(*dist_utils.gather_records(engine.sweep(can_shrink)))
    folded = engine.fold(*dist_utils.gather_records(engine.finalize(folded, None, use_prior=False)))
    if defer is not None and can_shrink:
        defer.append(folded)
        return None
    return prefix_rule_finish(engine, folded, n, batch, can_shrink, stats)


def prefix_rule_finish(engine, folded, n, batch, can_shrink, stats=None):
    """The host's part of :func:`prefix_rule`: one 64-byte record -> ``c_max`` (and, for
    ``can_shrink=False`` after a failure, the pass that keeps the later batches' previous state)."""
    row = _host_words(folded)
    star = (dist_utils.u64(row[_hip.R_FAIL_V]), row[_hip.R_FAIL_I])
    below = row[_hip.R_BELOW]
    last_safe = (dist_utils.u64(row[_hip.R_LAST_V]), row[_hip.R_LAST_I])
    max_key = (dist_utils.u64(row[_hip.R_MAX_V]), row[_hip.R_MAX_I])
    safe = row[_hip.R_SAFE]
    failed = star != _KEY_NONE

    keep = None
    if failed and not can_shrink:
        # cells after the batch that contains the first failure keep their previous state
        # (lyapunov.py:585-587 never touches later batches): key_keep = the key at sorted position
        # (below // batch + 1) * batch, computed AND consumed on the device
        keep = select_kth(engine, -1, batch, folded, n)
        folded = engine.fold(*dist_utils.gather_records(engine.finalize(folded, keep, use_prior=True)))
        if stats is not None:
            safe = _host_words(folded)[_hip.R_SAFE]
    if stats is not None:
        stats['safe'] = safe
        stats['below'] = below
        stats['folded'], stats['keep'] = folded, keep      # (device: key* and key_keep of the update)

    # c_max = values[order[max_index]], max_index as in lyapunov.py:590
    if failed:
        c_key = last_safe if below > 0 else max_key
    else:
        last_batch_start = ((n - 1) // batch) * batch
        if last_batch_start > 0:
            state = _host_words(select_kth(engine, last_batch_start - 1, batch, folded, n))
            c_key = (dist_utils.u64(state[_hip.S_KEY_V]), state[_hip.S_KEY_I])
        else:
            c_key = max_key
    return vbits_to_float(c_key[0])


class _HipAdaptiveEngine(object):
    """This rank's part of the adaptive branch: HIP kernels on device buffers (``sl_adaptive.hip``)."""

    def __init__(self, lyap, can_shrink):
        import torch
        self.lyap, self.can_shrink = lyap, can_shrink
        self.dev = lyap._ctx.torch_device
        self.torch = torch
        self.counts = torch.zeros(_hip.SORT_COUNT_WORDS, dtype=torch.int32, device=self.dev)
        self.refinement = None

    def _empty(self, *shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.int64, device=self.dev)

    def pack(self):
        """One row per cell of the shard: ``[vbits(V), index, decrease, threshold(tau=1), N, flags]``
        from one sweep with ``tau = 1`` (``threshold(x, tau') = threshold(x, 1) tau'`` exactly)."""
        ly, torch = self.lyap, self.torch
        d = ly.discretization.ndim
        count = ly._hi - ly._lo
        records = torch.empty((max(count, 1), 2 + 2 * d), dtype=torch.float64, device=self.dev)
        scratch_bits = torch.empty_like(ly._d_neg)
        values = ly._d_values                  # (materialises V with the model as it is)
        ly._builder.upload(ly.policy, ly.dynamics, ly.lyapunov_function, ly._lipschitz_lyapunov,
                           ly._lipschitz_dynamics, 1.0)
        try:
            ly._ctx.lyap_sweep(ly._lo, ly._hi, ly._d_init, values, scratch_bits, ly._d_result, records)
        finally:
            ly._upload_model()
        rows = self._empty(max(count, 1), _hip.ADAPTIVE_ROW_WORDS)
        prior_bits = None if self.can_shrink else ly._d_safe
        prior_ref = None if self.can_shrink else ly._refinement_shard()
        ly._ctx.adaptive_pack(ly._lo, ly._hi, values, records, 2 + 2 * d, ly._d_init,
                              prior_bits, prior_ref, rows)
        return rows[:count]

    def splitters(self, positions, n):
        """Keys at the given sorted positions (device select states, one row each)."""
        ly = self.lyap
        shard = _HipShardEngine(ly)
        states = self._empty(max(len(positions), 1), _hip.SELECT_WORDS)
        # one multi-k select: its states are rows of one block
        for r, state in enumerate(select_kth_many(shard, list(positions), 1, ly._d_folded, n)):
            states[r].copy_(state)
        return states[:len(positions)]

    def partition(self, rows, splitters):
        """Rows ordered by the rank that owns their sorted position -> ``(rows, counts per rank)``."""
        ly = self.lyap
        count = len(rows)
        dest = self._empty(max(count, 1), dtype=self.torch.uint8)
        ly._ctx.adaptive_dest(count, rows, splitters, len(splitters), dest)
        perm = self._empty(max(count, 1))
        buckets = self._empty(256)
        ly._ctx.partition_by_digit(count, dest, perm, buckets, self.counts)
        out = self._empty(max(count, 1), _hip.ADAPTIVE_ROW_WORDS)
        ly._ctx.gather_rows(count, _hip.ADAPTIVE_ROW_WORDS, perm, rows, out)
        return out[:count], buckets

    def sort(self, rows):
        """``order[q]`` = row at this rank's q-th sorted position, and the sorted value bits."""
        ly = self.lyap
        m = len(rows)
        keys, vals = self._empty(max(m, 1)), self._empty(max(m, 1))
        ly._ctx.adaptive_sort_keys(m, rows, keys, vals)
        ly._ctx.sort_pairs(m, keys, vals, self._empty(max(m, 1)), self._empty(max(m, 1)), self.counts)
        return vals[:m], keys[:m]

    def analyse(self, rows, order, pos0, batch, max_refinement, safety_factor):
        ly = self.lyap
        m = len(rows)
        info = self._empty(max(-(-m // batch), 1), 4, dtype=self.torch.int32)
        first_break = self._empty(1)
        ly._ctx.adaptive_analyse(m, pos0, batch, rows, order, ly.tau, safety_factor, max_refinement,
                                 info, first_break)
        return info, first_break

    def apply(self, rows, order, info, pos0, batch, b_star, safety_factor):
        ly = self.lyap
        m = len(rows)
        out = self._empty(max(m, 1), 3)
        ly._ctx.adaptive_apply(m, pos0, batch, rows, order, info, ly.tau, safety_factor, b_star, out)
        return out[:m]

    def scatter(self, out_rows):
        """Returned rows -> the shard's mask words and refinement array; -> safe cells (device)."""
        ly = self.lyap
        count = ly._hi - ly._lo
        self.refinement = self.torch.zeros(max(count, 1), dtype=self.torch.int64, device=self.dev)[:count]
        safe_count = self._empty(1)
        ly._ctx.adaptive_scatter(ly._lo, ly._hi, len(out_rows), out_rows, ly._d_init, ly._d_safe,
                                 self.refinement, safe_count)
        return safe_count


def _order_int(vbits):
    """uint64 value bits -> int64 with the same order (for MAX all-reduces of int64 tensors)."""
    v = (int(vbits) & _U64_MAX) ^ (1 << 63)
    return v - (1 << 64) if v >= (1 << 63) else v


def adaptive_rule(engine, n, batch, max_refinement, safety_factor, stats=None):
    """The adaptive branch of ``lyapunov.py:540-582`` over sharded cells; returns ``c_max``.

    ``engine`` (HIP in production, NumPy in the CPU tests of the multi-rank path) provides the
    primitives; this function is the exchange: rank r owns the sorted positions ``[r per, (r+1) per)``
    (``per`` a whole number of batches), rows travel there and back with one all-to-all each, the
    first batch that ends the loop is a MIN all-reduce, ``c_max`` two small all-reduces."""
    import torch
    rank, world = dist_utils.rank_and_world()
    per = max(-(-(-(-n // world)) // batch) * batch, batch)
    pos0 = rank * per
    rows = engine.pack()
    send_counts = None
    if world > 1:
        splitters = engine.splitters([r * per for r in range(1, world)], n)
        rows, buckets = engine.partition(rows, splitters)
        send_counts = [int(v) for v in buckets[:world].cpu()]
        rows, recv_counts = dist_utils.exchange_rows(rows, send_counts)
    m = len(rows)
    order, sorted_bits = engine.sort(rows)
    info, first_break = engine.analyse(rows, order, pos0, batch, max_refinement, safety_factor)
    b_star = int(dist_utils.allreduce_min_(first_break)[0])
    out = engine.apply(rows, order, info, pos0, batch, b_star, safety_factor)
    if world > 1:
        out, _ = dist_utils.exchange_rows(out, recv_counts, send_counts)
    safe = dist_utils.allreduce_sum_(engine.scatter(out))

    # c_max = values[order[i + bound + refine_bound - 1]], i = start of the last batch the loop
    # touched (lyapunov.py:590): its owner publishes bound and refine_bound, then the owner of that
    # sorted position publishes the value bits (position -1 reads the LARGEST value, as in NumPy)
    nbatches = -(-n // batch)
    b_last = b_star if b_star < nbatches else nbatches - 1
    mine = pos0 <= b_last * batch < pos0 + m
    pair = torch.zeros(2, dtype=torch.int64, device=first_break.device)
    if mine:
        pair += info[b_last - pos0 // batch][[1, 3]].to(torch.int64)
    pair = dist_utils.allreduce_sum_(pair).cpu()
    position = b_last * batch + int(pair[0]) + int(pair[1]) - 1
    word = torch.full((1,), -(1 << 63), dtype=torch.int64, device=first_break.device)
    if position < 0:
        if m:
            word[0] = _order_int(int(sorted_bits[m - 1]))
    elif pos0 <= position < pos0 + m:
        word[0] = _order_int(int(sorted_bits[position - pos0]))
    word = int(dist_utils.allreduce_max_(word)[0])
    if stats is not None:
        stats['safe'] = int(safe[0])
        stats['b_star'] = b_star
    return vbits_to_float((word & _U64_MAX) ^ (1 << 63))


class _CMaxView(dict):
    """``lyapunov.feed_dict[lyapunov.c_max]`` of the notebooks keeps working: any key reads c_max."""

    def __init__(self, owner):
        super(_CMaxView, self).__init__()
        self._owner = owner

    def __getitem__(self, key):
        return self._owner.c_max


def _unique_rows(array):
    """Unique rows in byte-wise order (``safe_learning/utilities.py:496-516``)."""
    array = np.ascontiguousarray(array)
    void = np.dtype((np.void, array.dtype.itemsize * array.shape[1]))
    _, keep = np.unique(array.view(void), return_index=True)
    return array[keep]


def perturb_actions(states, actions, perturbations, limits=None):
    """State-action pairs around a baseline policy (``lyapunov.py:609-651``): every state is
    paired with ``action + perturbation`` for every perturbation row; with ``limits`` the actions
    are clipped and duplicate rows dropped."""
    states = np.asarray(states, dtype=np.float64)
    perturbations = np.atleast_2d(np.asarray(perturbations, dtype=np.float64))
    count, state_dim = len(perturbations), states.shape[1]
    pairs = np.column_stack((np.repeat(states, count, axis=0),
                             np.repeat(actions, count, axis=0)
                             + np.tile(perturbations, (len(states), 1))))
    if limits is not None:
        limits = np.asarray(limits, dtype=np.float64)
        np.clip(pairs[:, state_dim:], limits[:, 0], limits[:, 1], out=pairs[:, state_dim:])
        pairs = _unique_rows(pairs)
    return pairs


def _safe_words_device(lyapunov):
    """Mask words of the WHOLE grid on the device (every rank holds them after ``update_safe_set``;
    a host array the caller edited is packed first)."""
    import torch
    n = lyapunov.discretization.nindex
    dev = lyapunov._ctx.torch_device
    if not lyapunov._safe_device_current() or (lyapunov._collective and lyapunov._d_safe_full is None):
        lyapunov.mask_uploads += 1
        d_bytes = torch.from_numpy(lyapunov.safe_set.view(np.uint8)).to(dev)
        words = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        lyapunov._ctx.bytes_to_bits(n, d_bytes, words)
        return words
    return lyapunov._d_safe_full if lyapunov._d_safe_full is not None else lyapunov._d_safe


class _SampleKernels(object):
    """Device steps of ``get_safe_sample`` (``sl_sample.hip``, ``sl_partition_by_digit``,
    ``sl_sort_pairs``) on the engine context of a Lyapunov object."""

    def __init__(self, lyapunov):
        import torch
        self.torch, self.ctx, self.dev = torch, lyapunov._ctx, lyapunov._ctx.torch_device
        self.counts = torch.zeros(_hip.SORT_COUNT_WORDS, dtype=torch.int32, device=self.dev)

    def _empty(self, *shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.int64, device=self.dev)

    def safe_indices(self, words, n):
        """Flat indices of the set bits, ascending (``np.where(safe_set)``, lyapunov.py:729): counted
        and written from the mask words (``sl_bits_count`` / ``sl_bits_to_indices``: n / 8 bytes read,
        8 bytes per safe cell written - not a partition of all n cells)."""
        nblocks = -(-(-(-n // 64)) // 256)
        block_counts = self._empty(max(nblocks, 1), dtype=self.torch.int32)
        offsets = self._empty(nblocks + 1)
        count = self.ctx.bits_count(n, words, block_counts, offsets)
        out = self._empty(max(count, 1))[:count]
        self.ctx.bits_to_indices(n, words, offsets, out)
        return out

    def take(self, rows, picks):
        """``rows[picks]`` for a 1-D or 2-D int64 / float64 tensor (``sl_gather_rows``)."""
        words = 1 if rows.dim() == 1 else rows.shape[1]
        out = self.torch.empty((len(picks),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.dev)
        self.ctx.gather_rows(len(picks), words, picks, rows.contiguous(), out)
        return out

    def states(self, indices):
        d = self._d
        out = self._empty(max(len(indices), 1), d, dtype=self.torch.float64)[:len(indices)]
        self.ctx.index_to_state(len(indices), indices.contiguous(), out)
        return out

    def pairs(self, states, actions, perturbations, limits):
        """``perturb_actions`` (lyapunov.py:609-651): every state with every perturbed, clipped
        action; with limits, duplicate rows dropped in the byte-wise order of ``unique_rows``."""
        torch = self.torch
        pert = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(perturbations), dtype=np.float64)).to(self.dev)
        lim = None
        if limits is not None:
            lim = torch.from_numpy(np.ascontiguousarray(limits, dtype=np.float64)).to(self.dev)
        count, d, m = len(states), states.shape[1], actions.shape[1]
        total, width = count * len(pert), d + m
        rows = self._empty(max(total, 1), width, dtype=torch.float64)[:total]
        self.ctx.perturb_pairs(count, d, m, states.contiguous(), actions.contiguous(), len(pert), pert,
                               lim, rows)
        if limits is None or total == 0:
            return rows
        # np.unique of the rows viewed as raw bytes: stable radix sorts, last column first
        order = torch.arange(total, dtype=torch.int64, device=self.dev)
        keys, tmp_k, tmp_v = self._empty(total), self._empty(total), self._empty(total)
        for column in range(width - 1, -1, -1):
            self.ctx.rows_sort_key(total, width, column, rows, order, keys)
            self.ctx.sort_pairs(total, keys, order, tmp_k, tmp_v, self.counts)
        flags = self._empty(total, dtype=torch.uint8)
        self.ctx.rows_duplicate_flags(total, width, rows, order, flags)
        perm, buckets = self._empty(total), self._empty(256)
        self.ctx.partition_by_digit(total, flags, perm, buckets, self.counts)     # first of each group first
        keep = self.take(order, perm[:int(buckets[0])])
        return self.take(rows, keep)


def get_safe_sample(lyapunov, perturbations=None, limits=None, positive=False,
                    num_samples=None, actions=None):
    """Most uncertain safe state-action pair for the next measurement (``lyapunov.py:657-797``).

    Everything that scales with the safe set runs in HIP kernels on the device: the safe cells are
    compacted from the bit mask (stable partition), their states / perturbed and clipped actions /
    duplicate removal (radix sorts in ``unique_rows``' byte order) are ``sl_sample.hip`` kernels,
    the GP posterior, ``V`` and ``L_v`` at the candidates come from the explicit-point entry, the
    membership test ``safe_set[state_to_index(mean)]`` and the arg-max are kernels too.  Only the
    winning row travels to the host.  Returns ``(state_action[1, d+m], bound)``."""
    import warnings
    import torch
    from . import _evaluate
    grid = lyapunov.discretization
    d, n = grid.ndim, grid.nindex
    lyapunov._upload_model()                 # the kernels below read the grid of the engine's model
    kernels = _SampleKernels(lyapunov)
    kernels._d = d
    words = _safe_words_device(lyapunov)
    safe_idx = kernels.safe_indices(words, n)
    if num_samples is not None and len(safe_idx) > num_samples:
        pick = np.random.choice(len(safe_idx), num_samples, replace=True)     # the reference's draw
        safe_idx = kernels.take(safe_idx, torch.from_numpy(pick.astype(np.int64)).to(safe_idx.device))
    safe_states = kernels.states(safe_idx)
    safe_actions = None
    if perturbations is None:
        # lyapunov.py:737-741: every safe state with every row of ``actions`` (a meshgrid of the two
        # RAVELLED arrays, i.e. one state and one action dimension - the 1-D examples)
        grid_actions = np.asarray(actions, dtype=np.float64)
        if d == 1:
            # (the reference ravels `actions` whatever its shape, so every element is one action)
            # state-major pairs on the device: the pairs of perturb_actions around a zero baseline
            # (0 + a = a exactly), without limits - no clipping, no duplicate removal
            zero = torch.zeros((len(safe_states), 1), dtype=torch.float64, device=safe_states.device)
            state_actions = kernels.pairs(safe_states, zero, grid_actions.reshape(-1, 1), None)
        else:
            host_states = safe_states.cpu().numpy()
            mesh = np.meshgrid(host_states, actions, indexing='ij')
            state_actions = torch.from_numpy(np.column_stack([m.ravel() for m in mesh])).to(safe_states.device)
    else:
        safe_actions = _evaluate.policy(lyapunov.policy, safe_states)
        state_actions = kernels.pairs(safe_states, safe_actions, perturbations, limits)

    def evaluate(pairs):
        count = len(pairs)
        mean, std = _evaluate.dynamics(lyapunov.dynamics, pairs[:, :d].contiguous(),
                                       pairs[:, d:].contiguous())
        value, lv = _evaluate.value(lyapunov.lyapunov_function, mean.contiguous(),
                                    lyapunov._lipschitz_lyapunov)
        bound = torch.empty(max(count, 1), dtype=torch.float64, device=pairs.device)[:count]
        inside = torch.empty(max(count, 1), dtype=torch.uint8, device=pairs.device)[:count]
        lyapunov._ctx.sample_bounds(count, d, lv.shape[1], std.contiguous(), lv.contiguous(),
                                    value.contiguous(), lyapunov.c_max, bound, inside)
        return mean.contiguous(), bound, inside

    def best_of(bound, mask):
        out = torch.empty(2, dtype=torch.int64, device=bound.device)
        lyapunov._ctx.argmax_masked(len(bound), bound, mask, out)
        index, seen = (int(v) for v in out.cpu())
        return index, seen

    mean, bound, maps_inside = evaluate(state_actions)
    if not positive:
        lyapunov._ctx.state_membership(len(mean), mean, words, maps_inside)
    best, seen = best_of(bound, maps_inside)
    if seen == 0:
        warnings.warn("No safe state-action pairs found! Using backup policy ...", RuntimeWarning)
        state_actions = kernels.pairs(safe_states, safe_actions, np.array([[0.]]), limits)
        _, bound, _ = evaluate(state_actions)
        best, _ = best_of(bound, None)
    return state_actions[[best]].cpu().numpy(), float(bound[best])


def get_lyapunov_region(lyapunov, discretization, init_node):
    """Region around ``init_node`` that the reference's priority-queue flood of the function values
    visits before it reaches the grid boundary or has to descend (``lyapunov.py:59-139``).

    The values come from the engine's value pass; the flood itself is computed on the GPU in
    parallel form (``sl_lyapunov_region``, ``csrc/sl_region.hip``): the minimax distance from the
    start node as a relaxation fixpoint, the stop level, the last regular pop and the node it
    descends to.  Equal to the reference's region wherever the touched cells have distinct values
    (``tests/golden/reference_regions.npz``, computed by the reference's own function); a start node
    on the grid boundary gives the empty region (the reference does so on an upper boundary; on a
    lower one it fails to notice - ``0 == tuple`` - and wraps around the grid)."""
    import torch
    helper = Lyapunov.__new__(Lyapunov)
    helper._bare_init(discretization, lyapunov)
    shape = tuple(int(v) for v in discretization.num_points)
    start = int(np.ravel_multi_index(tuple(int(v) for v in init_node), shape))
    n = discretization.nindex
    dev = helper._ctx.torch_device
    values = helper.gather_values().contiguous()
    work = torch.empty(n, dtype=torch.float64, device=dev)
    region = torch.empty(n, dtype=torch.uint8, device=dev)
    helper._upload_model()
    helper._ctx.lyapunov_region(values, start, work, region)
    return region.cpu().numpy().astype(bool).reshape(shape)
