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
        self.c_max = 0.
        self.feed_dict = _CMaxView(self)
        self.initial_safe_set = initial_set
        # safe_set starts as the initial set (lyapunov.py:187-192)
        self._safe_host = np.zeros(self.discretization.nindex, dtype=bool)
        if initial_set is not None:
            self._safe_host[initial_set] = True
        self._safe_host_valid = True
        self._safe_dev_valid = False
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
        self._d_values = torch.zeros(cap, dtype=torch.float64, device=dev)
        self._d_init = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_neg = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_safe = torch.zeros(wcap, dtype=torch.int64, device=dev)
        self._d_result = torch.zeros(_hip.RESULT_WORDS, dtype=torch.int64, device=dev)
        self._d_hist = torch.zeros(256, dtype=torch.int64, device=dev)
        self._values_host = None
        self._d_values_full = None      # all shards of V: gathered on demand (gather_values)
        self._d_safe_full = None        # all shards' mask words, gathered by update_safe_set
        self._init_version = None
        self._init_object = None

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

    # ---- reference attribute surface -----------------------------------------------------
    @property
    def initial_safe_set(self):
        return self._initial_safe_set

    @initial_safe_set.setter
    def initial_safe_set(self, value):
        self._initial_safe_set = value
        self._init_version = None

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
        needs nothing else); the first read after ``update_values`` gathers the shards and is
        therefore COLLECTIVE: read it on every rank (or call :meth:`gather_values` on every
        rank first), never on one rank only."""
        if self._values_host is None:
            self._values_host = self.gather_values().cpu().numpy()
        return self._values_host

    def gather_values(self):
        """All shards of V as one device tensor ``float64[nindex]`` (collective when the grid is
        sharded; cached until the next ``update_values``).  One ``all_gather_into_tensor`` into a
        pre-sized buffer - 2.1 GB per GPU at 128^4, which is why it is not done eagerly."""
        if not self._collective:
            return self._d_values[:self._hi - self._lo]
        if self._d_values_full is None:
            self._d_values_full = dist_utils.allgather_equal(self._d_values,
                                                             self.discretization.nindex)
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
        A local read on every rank (``update_safe_set`` gathers the shards' mask words)."""
        if not self._safe_host_valid:
            import torch
            n = self.discretization.nindex
            words = self._d_safe_full if self._d_safe_full is not None else self._d_safe
            d_bytes = torch.empty(max(-(-n // 8) * 8, 8), dtype=torch.uint8,
                                  device=self._ctx.torch_device)
            self._ctx.bits_to_bytes(n, words, d_bytes)
            self._safe_host[:] = d_bytes[:n].cpu().numpy().astype(bool)
            self._safe_host_valid = True
            # the host array may now be edited by the caller: it becomes the truth again
            self._safe_dev_valid = False
        return self._safe_host

    def _safe_bytes_device(self):
        """The safe set as ``uint8[nindex]`` on the device (no host copy): the mask words every
        rank holds after ``update_safe_set``, or the host array if the caller edited it."""
        import torch
        n = self.discretization.nindex
        dev = self._ctx.torch_device
        if self._safe_host_valid and not self._safe_dev_valid:
            return torch.from_numpy(self._safe_host.view(np.uint8)).to(dev)
        words = self._d_safe_full if self._d_safe_full is not None else self._d_safe
        d_bytes = torch.empty(max(-(-n // 8) * 8, 8), dtype=torch.uint8, device=dev)
        self._ctx.bits_to_bytes(n, words, d_bytes)
        return d_bytes[:n]

    @safe_set.setter
    def safe_set(self, value):
        self._safe_host[:] = value
        self._safe_host_valid = True
        self._safe_dev_valid = False

    @property
    def _refinement(self):
        """Refinement N(x) per cell (``lyapunov.py:220-225``): the array kept by the adaptive
        branch, otherwise 1 on safe cells and 0 elsewhere (``:531, 586, 601-606``)."""
        if self._refinement_host is None and getattr(self, '_refinement_dev', None) is not None:
            self._refinement_host = self._refinement_dev.cpu().numpy().astype(int)
        if self._refinement_host is not None:
            return self._refinement_host
        return self.safe_set.astype(int)

    def is_safe(self, state):
        """``lyapunov.py:290-303``."""
        return self.safe_set[self.discretization.state_to_index(state)]

    def safety_constraint(self, policy, include_initial=True):
        """``bool[nindex]``: where the decrease condition holds under ``policy`` - a per-vertex
        action array ``[nindex, m]`` or a policy spec (``lyapunov.py:378-406``).  The reference
        method cannot run as written (it compares with the bound method ``self.threshold`` and
        hands the discretization object to the dynamics); this is what its docstring describes,
        evaluated by the same sweep kernel as ``update_safe_set``."""
        import torch
        own_policy = self.policy
        self.policy = policy
        try:
            self._upload_model()
            self._refresh_init_bits()
            d_neg = torch.zeros_like(self._d_neg)
            self._ctx.lyap_sweep(self._lo, self._hi, self._d_init, self._d_values, d_neg,
                                 self._d_result)
        finally:
            self.policy = own_policy
            self._upload_model()
        count = self._hi - self._lo
        d_bytes = torch.empty(max(-(-count // 8) * 8, 8), dtype=torch.uint8,
                              device=self._ctx.torch_device)
        self._ctx.bits_to_bytes(count, d_neg, d_bytes)
        sizes = [self._bounds[r + 1] - self._bounds[r] for r in range(self._world)]
        mask = dist_utils.allgather_concat(d_bytes[:count], sizes).cpu().numpy().astype(bool)
        if include_initial and self._initial_safe_set is not None:
            mask[self._initial_safe_set] = True
        return mask

    # ---- device helpers ------------------------------------------------------------------
    def _upload_mask(self, host_mask, d_bits):
        """bool[nindex] host mask -> this rank's bit words."""
        import torch
        count = self._hi - self._lo
        if count == 0:
            return
        chunk = np.ascontiguousarray(host_mask[self._lo:self._hi]).view(np.uint8)
        d_bytes = torch.from_numpy(chunk).to(self._ctx.torch_device)
        self._ctx.bytes_to_bits(count, d_bytes, d_bits)

    def _refresh_init_bits(self):
        init = self._initial_safe_set
        if init is None:
            version = 'none'
        else:
            arr = np.asarray(init)
            # large masks are identified by object identity (re-assign ``initial_safe_set`` after
            # editing one in place); small ones also by content
            checksum = int(np.count_nonzero(arr)) if arr.size <= (1 << 20) else -1
            version = (arr.shape, arr.dtype.str, checksum)
        # the cached object is held strongly and compared by identity: an id() could be recycled
        if version == self._init_version and init is self._init_object:
            return
        if init is None:
            self._d_init.zero_()
        else:
            arr = np.asarray(init)
            if arr.dtype == bool and arr.shape == (self.discretization.nindex,):
                mask = arr
            else:
                mask = np.zeros(self.discretization.nindex, dtype=bool)
                mask[init] = True
            self._upload_mask(mask, self._d_init)
        self._init_version = version
        self._init_object = init

    # ---- reference methods ---------------------------------------------------------------
    def update_values(self):
        """Recompute V on the grid (``lyapunov.py:305-322``)."""
        self._upload_model()
        self._ctx.values(self._lo, self._hi, self._d_values)
        self._values_host = None
        self._d_values_full = None       # shards only; ``values`` / ``gather_values`` gather lazily

    def update_safe_set(self, can_shrink=True, max_refinement=1, safety_factor=1.,
                        parallel_iterations=1):
        """Recompute the safe set (``lyapunov.py:407-606``).  ``max_refinement > 1`` on an adaptive
        instance takes the adaptive branch (``:540-582``): the per-cell quantities still come from
        one GPU sweep, the sequential refinement bookkeeping runs on the host like the reference's."""
        if self.adaptive and max_refinement > 1:
            return self._update_safe_set_adaptive(can_shrink, max_refinement, safety_factor)
        self._refinement_host = None
        self._refinement_dev = None
        self._upload_model()
        self._refresh_init_bits()
        if not can_shrink and not self._safe_dev_valid:          # lyapunov.py:507-510
            self._upload_mask(self._safe_host, self._d_safe)
        engine = _HipShardEngine(self)
        stats = {}
        self.c_max = prefix_rule(engine, self.discretization.nindex, int(config.gp_batch_size),
                                 can_shrink, self._ctx.torch_device, stats)
        self.safe_count = stats['safe']       # cells in the safe set (all ranks), no mask copy
        self._safe_host_valid = False
        self._safe_dev_valid = True
        if self._collective:
            # shards start at multiples of 64 cells, so their mask words concatenate (4 MB per
            # rank at 128^4 over 8 GPUs); afterwards ``safe_set`` is a local read on every rank
            self._d_safe_full = dist_utils.allgather_equal(
                self._d_safe, -(-self.discretization.nindex // 64), out=self._safe_full_buffer())


    def _threshold_base_device(self):
        """``-|L_v(x)|_1 (1 + L_f(x))`` at every grid cell as a device tensor (the refined
        threshold of the adaptive branch is this times ``tau / N(x)``): ``threshold(x, tau=1)``
        with the same operation order, ``L_v`` from the point-evaluation kernels."""
        import torch
        from . import _evaluate
        n = self.discretization.nindex
        dev = self._ctx.torch_device
        lvs, lfs = self._lipschitz_lyapunov, self._lipschitz_dynamics
        states = None
        if not (np.isscalar(lvs) and np.isscalar(lfs)):
            states = _index_to_state_device(self.discretization,
                                            torch.arange(n, dtype=torch.int64, device=dev))
        if np.isscalar(lvs):
            lv = torch.full((n,), float(lvs), dtype=torch.float64, device=dev)
        else:
            lv = _evaluate.value(self.lyapunov_function, states, lvs)[1]
            if lv.shape[1] > 1:
                acc = lv[:, 0].abs()
                for k in range(1, lv.shape[1]):
                    acc = acc + lv[:, k].abs()
                lv = acc
            else:
                lv = lv[:, 0]
        if np.isscalar(lfs):
            lf = float(lfs)
        else:
            rows = torch.from_numpy(np.asarray(lfs.fun.matrix, dtype=np.float64)).to(dev)
            acc = None
            for r in range(rows.shape[0]):
                t = states[:, 0] * rows[r, 0]
                for k in range(1, states.shape[1]):
                    t = t + states[:, k] * rows[r, k]
                acc = t.abs() if acc is None else acc + t.abs()
            lf = lfs.constant + acc
        return (-lv) * (1. + lf) * 1.0

    def _update_safe_set_adaptive(self, can_shrink, max_refinement, safety_factor):
        """Adaptive discretisation (``lyapunov.py:445-487, 540-582``), bug-compatible: as written
        in the reference the refined check compares the decrease of every cell of the candidate
        run with each cell's refined threshold ``threshold(x, tau / N(x))`` (the refined points
        themselves are never evaluated), see DESIGN.md.

        Everything of grid size stays on the GPU: the per-cell quantities come from one sweep of
        this rank's shard, the ascending-V order from a stable device sort (no host ``argsort``),
        and the batch loop of the reference runs over device slices with a few scalar read-backs
        per batch; it ends at the first batch that cannot be refined, like the reference's."""
        import torch
        n, d = self.discretization.nindex, self.discretization.ndim
        dev = self._ctx.torch_device
        batch = int(config.gp_batch_size)
        safety_factor = max(float(safety_factor), 1.)
        self._upload_model()
        self._refresh_init_bits()
        lo, hi = self._lo, self._hi
        count = hi - lo
        sizes = [self._bounds[r + 1] - self._bounds[r] for r in range(self._world)]
        records = torch.empty((max(count, 1), 2 + 2 * d), dtype=torch.float64, device=dev)
        self._ctx.lyap_sweep(lo, hi, self._d_init, self._d_values, self._d_neg, self._d_result, records)
        decrease = dist_utils.allgather_concat(records[:count, 0].contiguous(), sizes)
        threshold = dist_utils.allgather_concat(records[:count, 1].contiguous(), sizes)
        d_bytes = torch.empty(max(-(-count // 8) * 8, 8), dtype=torch.uint8, device=dev)
        self._ctx.bits_to_bytes(count, self._d_neg, d_bytes)
        negative = dist_utils.allgather_concat(d_bytes[:count], sizes).to(torch.bool)
        values = self.gather_values()
        base = self._threshold_base_device()
        init_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        if self._initial_safe_set is not None:
            init = np.asarray(self._initial_safe_set)
            if init.dtype == bool and init.shape == (n,):
                init_mask = torch.from_numpy(init).to(dev)
            else:
                init_mask[torch.from_numpy(np.atleast_1d(init).astype(np.int64)).to(dev)] = True

        if can_shrink:
            safe_src = init_mask.clone()
            refine_src = init_mask.to(torch.int64)
        else:
            safe_src = self._safe_bytes_device().to(torch.bool).clone()
            refine_src = self._refinement_device().clone()
        order = torch.sort(values, stable=True).indices          # ascending (V, flat index)
        safe_sorted, refinement = safe_src[order], refine_src[order]
        ratio = safety_factor * threshold / decrease
        n_req_all = torch.ceil(torch.clamp(torch.where(torch.isnan(ratio), torch.zeros_like(ratio),
                                                       ratio), min=0.))
        n_req_all = torch.clamp(n_req_all, max=float(1 << 40)).to(torch.int64)   # inf -> "too many"

        start = bound = refine_bound = 0
        for start in range(0, n, batch):
            idx = order[start:start + batch]
            safe_b, ref_b = safe_sorted[start:start + batch], refinement[start:start + batch]
            neg_b = negative[idx]
            safe_b |= neg_b
            ref_b[neg_b] = 1
            unsafe = torch.nonzero(~safe_b, as_tuple=False)
            bound, refine_bound = (int(unsafe[0]) if len(unsafe) else 0), 0
            if not len(unsafe):
                continue
            ref_b[bound:] = n_req_all[idx[bound:]]
            ref_b[neg_b | init_mask[idx]] = 1
            checkable = ((ref_b >= 1) & (ref_b <= max_refinement))[bound:]
            stop = len(checkable) if bool(checkable.all()) else int(torch.argmin(checkable.to(torch.uint8)))
            if stop > 0:
                run = idx[bound:bound + stop]
                run_dec = decrease[run]
                refined_thr = base[run] * (self.tau / ref_b[bound:bound + stop].to(torch.float64))
                worst = float(run_dec.max()) if not bool(torch.isnan(run_dec).any()) else float('inf')
                refined_safe = worst < refined_thr
                refine_bound = stop if bool(refined_safe.all()) else int(torch.argmin(refined_safe.to(torch.uint8)))
                safe_b[bound:bound + refine_bound] = True
            if stop < len(checkable) or refine_bound < stop:
                safe_b[bound + refine_bound:] = False
                ref_b[bound + refine_bound:] = 0
                break

        self.c_max = float(values[order[start + bound + refine_bound - 1]])
        safe = torch.zeros(n, dtype=torch.bool, device=dev)
        safe[order[safe_sorted]] = True
        refine = torch.zeros(n, dtype=torch.int64, device=dev)
        refine[order] = refinement
        safe |= init_mask
        refine[init_mask] = 1
        self._refinement_dev = refine
        self._refinement_host = None
        self.safe_count = int(safe.sum())
        # the mask goes straight into this rank's bit words (and the gathered copy)
        full_bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        self._ctx.bytes_to_bits(n, safe.to(torch.uint8).contiguous(), full_bits)
        nwords = (count + 63) // 64
        self._d_safe[:nwords] = full_bits[lo // 64:lo // 64 + nwords]
        self._d_safe_full = full_bits if self._collective else None
        self._safe_host_valid = False
        self._safe_dev_valid = True

    def _refinement_device(self):
        """Refinement N(x) per cell as a device tensor (int64[nindex])."""
        import torch
        if getattr(self, '_refinement_dev', None) is not None:
            return self._refinement_dev
        if self._refinement_host is not None:
            return torch.from_numpy(np.asarray(self._refinement_host, dtype=np.int64)).to(
                self._ctx.torch_device)
        return self._safe_bytes_device().to(torch.int64)


class _HipShardEngine(object):
    """This rank's shard of the grid as the prefix rule sees it (all work in HIP kernels).

    ``sweep`` / ``finalize`` return the kernels' packed result record (``sl_sweep_result``, eight
    int64 words) as a DEVICE tensor; ``prefix_rule`` gathers the records of all ranks at once."""

    def __init__(self, lyap):
        self.lyap = lyap
        self.prior = None

    def sweep(self, can_shrink):
        """Decrease check of every cell; words R_FAIL_V / R_FAIL_I hold the local lexmin failing
        ``(vbits, index)``."""
        ly = self.lyap
        self.prior = ly._d_init if can_shrink else ly._d_safe.clone()     # lyapunov.py:500-510
        events = getattr(ly, 'sweep_events', None)
        if events is not None:                      # bench.py: HIP events on the kernel's stream
            import torch
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        ly._ctx.lyap_sweep(ly._lo, ly._hi, self.prior, ly._d_values, ly._d_neg, ly._d_result)
        if events is not None:
            stop.record()
            events.append((start, stop))
        return ly._d_result

    def finalize(self, star, keep, use_prior):
        """``safe = init | key < star | (prior & key >= keep)``; words R_BELOW, R_LAST_*, R_MAX_*
        hold the local statistics."""
        ly = self.lyap
        ly._ctx.lyap_finalize(ly._lo, ly._hi, ly._d_values, ly._d_init,
                              self.prior if use_prior else None, star, keep, ly._d_safe,
                              ly._d_result)
        return ly._d_result

    def select_hist(self, which, byte, prefix, vbits_equal):
        """Local 256-bin histogram of one radix-select pass (int64[256])."""
        ly = self.lyap
        ly._d_hist.zero_()
        ly._ctx.select_pass(ly._lo, ly._hi, ly._d_values, which, byte, prefix, vbits_equal,
                            ly._d_hist)
        return ly._d_hist


def select_kth(engine, k, device):
    """``(vbits, index)`` of the k-th smallest key (0-based) over all shards: two 8-pass radix
    selects (value bytes, then index bytes among equal values), one 2 KiB SUM all-reduce each."""
    def run(which, vbits_equal, rank_in):
        prefix, remaining = 0, rank_in
        for byte in range(7, -1, -1):
            hist = engine.select_hist(which, byte, prefix, vbits_equal)
            hist = dist_utils.allreduce_sum_(hist)
            hist = hist.cpu().numpy() if hasattr(hist, 'cpu') else np.asarray(hist)
            cum = np.cumsum(hist)
            digit = int(np.searchsorted(cum, remaining, side='right'))
            if digit > 0:
                remaining -= int(cum[digit - 1])
            prefix |= digit << (8 * byte)
        return prefix, remaining

    vbits, rank_among_equal = run(0, 0, k)
    index, _ = run(1, vbits, rank_among_equal)
    return vbits, index


def _reduce_key(rows, col_v, col_i, largest):
    keys = [(dist_utils.u64(r[col_v]), int(r[col_i])) for r in rows]
    return max(keys) if largest else min(keys)


def prefix_rule(engine, n, batch, can_shrink, device, stats=None):
    """The safe-set rule of ``lyapunov.py:512-606`` over sharded cells; returns ``c_max``.

    ``engine`` owns one contiguous shard (the HIP engine in production, a NumPy stand-in in the
    CPU tests of the multi-rank path).  Communication: ONE all-gather of the packed 64-byte result
    record after the sweep and ONE after the streaming pass (``distributed.gather_words``), each
    followed by one copy to the host; only ``can_shrink=False`` or the no-failure ``c_max`` quirk
    add sixteen 2 KiB SUM all-reduces of radix-select histograms.
    """
    rows = dist_utils.gather_words(engine.sweep(can_shrink))
    star = _reduce_key(rows, _hip.R_FAIL_V, _hip.R_FAIL_I, largest=False)
    rows = dist_utils.gather_words(engine.finalize(star, _KEY_NONE, use_prior=False))
    below = int(sum(int(r[_hip.R_BELOW]) for r in rows))
    last_safe = _reduce_key(rows, _hip.R_LAST_V, _hip.R_LAST_I, largest=True)
    max_key = _reduce_key(rows, _hip.R_MAX_V, _hip.R_MAX_I, largest=True)
    failed = star != _KEY_NONE

    if failed and not can_shrink:
        # cells after the batch that contains the first failure keep their previous state
        # (lyapunov.py:585-587 never touches later batches)
        end = (below // batch + 1) * batch
        keep = select_kth(engine, end, device) if end < n else _KEY_NONE
        record = engine.finalize(star, keep, use_prior=True)
        if stats is not None:
            rows = dist_utils.gather_words(record)
    if stats is not None:
        stats['safe'] = int(sum(int(r[_hip.R_SAFE]) for r in rows))
        stats['below'] = below

    # c_max = values[order[max_index]], max_index as in lyapunov.py:590
    if failed:
        c_key = last_safe if below > 0 else max_key
    else:
        last_batch_start = ((n - 1) // batch) * batch
        c_key = select_kth(engine, last_batch_start - 1, device) if last_batch_start > 0 else max_key
    return vbits_to_float(c_key[0])


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


def _index_to_state_device(grid, idx):
    """``GridWorld.index_to_state`` (functions.py:714-731) for a device tensor of flat indices."""
    import torch
    dev = idx.device
    rem, cols = idx, []
    for n in reversed([int(v) for v in grid.num_points]):
        cols.append(rem % n)
        rem = torch.div(rem, n, rounding_mode='floor')
    ijk = torch.stack(cols[::-1], dim=1).to(torch.float64)
    unit = torch.from_numpy(np.asarray(grid.unit_maxes, dtype=np.float64)).to(dev)
    offset = torch.from_numpy(np.asarray(grid.offset, dtype=np.float64)).to(dev)
    return ijk * unit + offset                                   # multiply, then add


def _state_to_index_device(grid, states):
    """``GridWorld.state_to_index`` (functions.py:733-752) on the device (round half to even)."""
    import torch
    dev = states.device
    lo = torch.from_numpy(np.asarray(grid.limits[:, 0], dtype=np.float64)).to(dev)
    hi = torch.from_numpy(np.asarray(grid.limits[:, 1], dtype=np.float64)).to(dev)
    inv = torch.from_numpy(1. / np.asarray(grid.unit_maxes, dtype=np.float64)).to(dev)
    offset = torch.from_numpy(np.asarray(grid.offset, dtype=np.float64)).to(dev)
    ijk = torch.round((torch.minimum(torch.maximum(states, lo), hi) - offset) * inv).to(torch.int64)
    flat = torch.zeros(len(states), dtype=torch.int64, device=dev)
    for k, n in enumerate(int(v) for v in grid.num_points):
        flat = flat * n + ijk[:, k]
    return flat


def _unique_rows_device(rows):
    """Unique rows in the byte-wise order of ``safe_learning/utilities.py:496-516`` (memcmp of the
    raw float64 bytes) on the device."""
    import torch
    n, k = rows.shape
    as_bytes = rows.contiguous().view(torch.uint8).reshape(n, 8 * k)
    return torch.unique(as_bytes, dim=0).contiguous().view(torch.float64).reshape(-1, k)


def get_safe_sample(lyapunov, perturbations=None, limits=None, positive=False,
                    num_samples=None, actions=None):
    """Most uncertain safe state-action pair for the next measurement (``lyapunov.py:657-797``).

    Everything that scales with the safe set stays on the GPU: the safe cells come from the
    device mask, their states / perturbed actions / duplicate removal are tensor operations, the
    GP posterior, ``V`` and ``L_v`` at the candidates come from the HIP kernels (explicit-point
    mode), the membership test ``safe_set[state_to_index(mean)]`` and the arg-max run on the
    device.  Only the winning row travels to the host.  Returns ``(state_action[1, d+m], bound)``."""
    import warnings
    import torch
    from . import _evaluate
    grid = lyapunov.discretization
    d = grid.ndim
    safe_bytes = lyapunov._safe_bytes_device()                   # uint8[nindex] on the device
    safe_idx = torch.nonzero(safe_bytes, as_tuple=False).reshape(-1)
    if num_samples is not None and len(safe_idx) > num_samples:
        pick = np.random.choice(len(safe_idx), num_samples, replace=True)     # the reference's draw
        safe_idx = safe_idx[torch.from_numpy(pick).to(safe_idx.device)]
    safe_states = _index_to_state_device(grid, safe_idx)
    safe_actions = None
    if perturbations is None:
        host_states = safe_states.cpu().numpy()
        mesh = np.meshgrid(host_states, actions, indexing='ij')
        state_actions = torch.from_numpy(np.column_stack([m.ravel() for m in mesh])).to(safe_states.device)
    else:
        safe_actions = _evaluate.policy(lyapunov.policy, safe_states)
        state_actions = _perturb_actions_device(safe_states, safe_actions, perturbations, limits)

    def evaluate(pairs):
        mean, std = _evaluate.dynamics(lyapunov.dynamics, pairs[:, :d].contiguous(),
                                       pairs[:, d:].contiguous())
        value, lv = _evaluate.value(lyapunov.lyapunov_function, mean.contiguous(),
                                    lyapunov._lipschitz_lyapunov)
        bound = std[:, [0]].clone()
        scaled = lv * std
        error = scaled[:, [0]].clone()
        for k in range(1, std.shape[1]):                         # left to right like the oracle
            bound = bound + std[:, [k]]
            error = error + scaled[:, [k]]
        return mean, bound, ((value + error) < lyapunov.c_max)[:, 0]

    mean, bound, maps_inside = evaluate(state_actions)
    if not positive:
        maps_inside &= safe_bytes[_state_to_index_device(grid, mean)].to(torch.bool)
    if not bool(maps_inside.any()):
        warnings.warn("No safe state-action pairs found! Using backup policy ...", RuntimeWarning)
        state_actions = _perturb_actions_device(safe_states, safe_actions, np.array([[0.]]), limits)
        _, bound, _ = evaluate(state_actions)
        best = int(torch.argmax(bound[:, 0]))
        return state_actions[[best]].cpu().numpy(), float(bound[best, 0])
    candidates = state_actions[maps_inside]
    bound = bound[maps_inside]
    best = int(torch.argmax(bound[:, 0]))
    return candidates[[best]].cpu().numpy(), float(bound[best, 0])


def _perturb_actions_device(states, actions, perturbations, limits):
    """``perturb_actions`` (``lyapunov.py:609-651``) on device tensors."""
    import torch
    dev = states.device
    pert = torch.from_numpy(np.atleast_2d(np.asarray(perturbations, dtype=np.float64))).to(dev)
    count, state_dim = len(pert), states.shape[1]
    acts = actions.repeat_interleave(count, dim=0) + pert.repeat(len(states), 1)
    if limits is not None:
        lim = torch.from_numpy(np.asarray(limits, dtype=np.float64)).to(dev)
        acts = torch.minimum(torch.maximum(acts, lim[:, 0]), lim[:, 1])
    pairs = torch.cat((states.repeat_interleave(count, dim=0), acts), dim=1)
    return _unique_rows_device(pairs) if limits is not None else pairs


def get_lyapunov_region(lyapunov, discretization, init_node):
    """Region around ``init_node`` in which the function ``lyapunov`` keeps increasing when the
    grid is flooded in order of increasing value (``lyapunov.py:59-139``).  The values come from
    the engine's value pass; the priority-queue flood fill is inherently sequential and runs on the
    host exactly like the reference's (same neighbour order and tie-breaking counter)."""
    import heapq
    import itertools
    helper = Lyapunov.__new__(Lyapunov)
    helper._bare_init(discretization, lyapunov)
    shape = tuple(int(v) for v in discretization.num_points)
    values = helper.values.reshape(shape)
    ndim = discretization.ndim
    steps = np.array(list(itertools.product((0, -1, 1), repeat=ndim))[1:])
    upper = np.array(shape) - 1
    start = tuple(int(v) for v in init_node)
    visited = np.zeros(shape, dtype=bool)
    visited[start] = True
    counter = itertools.count()
    heap = [(values[start], next(counter), start)]
    last = values[start]
    while heap:
        value, _, node = heapq.heappop(heap)
        at = np.array(node)
        if (at == 0).any() or (at == upper).any():
            visited[node] = False
            break
        if value < last:
            break
        last = value
        for step in steps:
            nb = tuple(int(v) for v in at + step)
            if not visited[nb]:
                visited[nb] = True
                heapq.heappush(heap, (values[nb], next(counter), nb))
    for _, _, node in heap:
        visited[node] = False
    return visited
