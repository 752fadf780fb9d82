"""NumPy stand-in for the HIP shard engine (TEST INFRASTRUCTURE).

Implements the primitives that ``safe_learning_amd.lyapunov.prefix_rule`` needs with the same
semantics as the kernels (sl_lyap_sweep / sl_fold_results / sl_lyap_finalize_dev / sl_select_begin /
sl_select_hist / sl_select_digit), on a shard of the grid, from the oracle's per-cell ``negative``
mask; records and select states are CPU int64 tensors with the kernels' word layout.  It lets the multi-rank orchestration
(sharding, key reductions, radix select with histogram all-reduce, c_max logic) run on CPU over
gloo."""

import numpy as np
import torch

U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)
KEY_NONE = (int(U64_MAX), (1 << 63) - 1)


def np_vbits(values):
    """Order-preserving float64 -> uint64 map of the kernels (sl_vbits): -0 == +0, NaN last."""
    v = np.where(values == 0, 0.0, values).astype(np.float64)
    raw = v.view(np.uint64)
    neg = (raw >> np.uint64(63)).astype(bool)
    out = np.where(neg, ~raw, raw | np.uint64(1 << 63))
    return np.where(np.isnan(v), U64_MAX, out).astype(np.uint64)


def _lex_less(vb, idx, key):
    kv, ki = np.uint64(key[0]), key[1]
    return (vb < kv) | ((vb == kv) & (idx < ki))


class NumpyShardEngine(object):
    def __init__(self, lo, hi, values, negative, init_mask, prev_safe):
        self.lo, self.hi = lo, hi
        self.vb = np_vbits(values[lo:hi])
        self.idx = np.arange(lo, hi, dtype=np.int64)
        self.negative = negative[lo:hi]
        self.init = init_mask[lo:hi]
        self.prev = prev_safe[lo:hi].copy()
        self.safe = np.zeros(hi - lo, dtype=bool)
        self.prior = None

    def _extreme(self, mask, largest):
        if not mask.any():
            return (0, -1) if largest else KEY_NONE
        vb, idx = self.vb[mask], self.idx[mask]
        order = np.lexsort((idx, vb))
        k = order[-1] if largest else order[0]
        return int(vb[k]), int(idx[k])

    @staticmethod
    def _record(fail=KEY_NONE, last=(0, -1), largest=(0, -1), below=0):
        """The kernels' packed result record (sl_sweep_result): eight int64 words."""
        words = np.array([fail[0], 0, last[0], 0, largest[0], 0, 0, 0], dtype=np.uint64).view(np.int64)
        words[1], words[3], words[5], words[6] = fail[1], last[1], largest[1], below
        return torch.from_numpy(words.copy())

    def sweep(self, can_shrink):
        self.prior = self.init if can_shrink else self.prev
        return self._record(fail=self._extreme(~(self.negative | self.prior), largest=False))

    @staticmethod
    def _key(words, at):
        return (int(words[at]) & 0xFFFFFFFFFFFFFFFF, int(words[at + 1]))

    def fold(self, records, count):
        """sl_fold_results: lexmin of fail, lexmax of last_safe / max_key, sums of the counters."""
        rows = records.numpy().reshape(count, 8)
        fail = min(self._key(r, 0) for r in rows)
        last = max(self._key(r, 2) for r in rows)
        largest = max(self._key(r, 4) for r in rows)
        out = self._record(fail=fail, last=last, largest=largest, below=int(rows[:, 6].sum()))
        out[7] = int(rows[:, 7].sum())
        return out

    def finalize(self, folded, keep_state, use_prior):
        """sl_lyap_finalize_dev: key* = folded.fail, key_keep = the select state's key."""
        star = self._key(folded.numpy(), 0)
        keep = KEY_NONE if keep_state is None else self._key(keep_state.numpy(), 2)
        below = _lex_less(self.vb, self.idx, star)
        self.safe = self.init | below
        if use_prior:
            self.safe |= self.prior & ~_lex_less(self.vb, self.idx, keep)
        out = self._record(fail=star, last=self._extreme(below, largest=True),
                           largest=self._extreme(np.ones_like(below), largest=True),
                           below=int(below.sum()))
        out[7] = int(self.safe.sum())
        return out

    def select_begin(self, k, batch, folded, n, slot=None):
        """sl_select_begin: [prefix, remaining, key.vbits, key.index, rank, none, 0, 0]."""
        rank = k if k >= 0 else (int(folded[6]) // batch + 1) * batch
        none = int(rank < 0 or rank >= n)
        words = np.array([0, 0, int(U64_MAX), 0, 0, 0, 0, 0], dtype=np.uint64).view(np.int64)
        words[1], words[3], words[4], words[5] = rank, KEY_NONE[1], rank, none
        return torch.from_numpy(words.copy())

    def select_hist(self, which, byte, state, slot=None):
        words = state.numpy()
        if words[5]:
            return torch.zeros(256, dtype=torch.int64)
        prefix = np.uint64(int(words[0]) & 0xFFFFFFFFFFFFFFFF)
        vbits_equal = np.uint64(int(words[2]) & 0xFFFFFFFFFFFFFFFF)
        key = self.vb if which == 0 else self.idx.astype(np.uint64)
        take = np.ones(len(key), dtype=bool) if which == 0 else (self.vb == vbits_equal)
        shift = np.uint64(8 * byte)
        if byte < 7:
            himask = np.uint64((0xFFFFFFFFFFFFFFFF << (8 * byte + 8)) & 0xFFFFFFFFFFFFFFFF)
            take &= (key & himask) == (prefix & himask)
        digits = ((key[take] >> shift) & np.uint64(0xFF)).astype(np.int64)
        return torch.from_numpy(np.bincount(digits, minlength=256).astype(np.int64))

    def stack_hists(self, hists):
        return torch.stack(hists)

    def select_digit(self, which, byte, hist, state):
        """sl_select_digit on the all-reduced histogram (in place on ``state``)."""
        words = state.numpy()
        if words[5]:
            return
        cum = np.cumsum(hist.numpy())
        digit = int(np.searchsorted(cum, int(words[1]), side="right"))
        if digit > 0:
            words[1] -= int(cum[digit - 1])
        prefix = (int(words[0]) & 0xFFFFFFFFFFFFFFFF) | (digit << (8 * byte))
        if byte == 0:
            words[2 if which == 0 else 3] = np.array([prefix], dtype=np.uint64).view(np.int64)[0]
            prefix = 0
        words[0] = np.array([prefix], dtype=np.uint64).view(np.int64)[0]


class NumpyAdaptiveEngine(object):
    """NumPy stand-in for ``safe_learning_amd.lyapunov._HipAdaptiveEngine`` (sl_adaptive.hip): the
    primitives of ``adaptive_rule`` on CPU tensors, from the oracle's per-cell decrease / threshold.
    Rows: [vbits, index, decrease bits, threshold(tau = 1) bits, refinement, flags]."""

    F_INIT, F_PRIOR = 1, 2

    def __init__(self, lo, hi, values, decrease, thr0, tau, init_mask, prior_safe, prior_ref, shard):
        self.lo, self.hi, self.tau = lo, hi, tau
        self.values, self.decrease, self.thr0 = values[lo:hi], decrease[lo:hi], thr0[lo:hi]
        self.init = init_mask[lo:hi]
        self.prior = self.init if prior_safe is None else prior_safe[lo:hi]
        self.prior_ref = (self.init.astype(np.int64) if prior_safe is None else
                          (self.prior.astype(np.int64) if prior_ref is None else prior_ref[lo:hi].astype(np.int64)))
        self.shard = shard                         # a NumpyShardEngine of the same cells (radix select)
        self.safe = self.refinement = None

    def pack(self):
        rows = np.empty((self.hi - self.lo, 6), dtype=np.int64)
        rows[:, 0] = np_vbits(self.values).view(np.int64)
        rows[:, 1] = np.arange(self.lo, self.hi)
        rows[:, 2] = np.ascontiguousarray(self.decrease, dtype=np.float64).view(np.int64)
        rows[:, 3] = np.ascontiguousarray(self.thr0, dtype=np.float64).view(np.int64)
        rows[:, 4] = self.prior_ref
        rows[:, 5] = self.init * self.F_INIT + self.prior * self.F_PRIOR
        return torch.from_numpy(rows)

    def splitters(self, positions, n):
        from safe_learning_amd.lyapunov import select_kth_many
        states = [state.clone() for state in select_kth_many(self.shard, list(positions), 1, None, n)]
        return torch.stack(states) if states else torch.zeros((0, 8), dtype=torch.int64)

    def partition(self, rows, splitters):
        r = rows.numpy()
        vb, idx = r[:, 0].view(np.uint64), r[:, 1]
        dest = np.zeros(len(r), dtype=np.int64)
        for s, state in enumerate(splitters.numpy()):
            key = (int(state[2]) & 0xFFFFFFFFFFFFFFFF, int(state[3]))
            dest[~_lex_less(vb, idx, key)] = s + 1
        perm = np.argsort(dest, kind="stable")
        buckets = np.bincount(dest, minlength=256)[:256].astype(np.int64)
        return torch.from_numpy(r[perm].copy()), torch.from_numpy(buckets)

    def sort(self, rows):
        vb = rows.numpy()[:, 0].view(np.uint64)
        order = np.argsort(vb, kind="stable")           # rows arrive in ascending index order
        return torch.from_numpy(order.astype(np.int64)), torch.from_numpy(vb[order].view(np.int64).copy())

    def _cells(self, rows, order, safety_factor):
        r = rows.numpy()[order.numpy()]
        dec, thr0 = r[:, 2].copy().view(np.float64), r[:, 3].copy().view(np.float64)
        thr = thr0 * self.tau
        neg = dec < thr
        init, prior = (r[:, 5] & self.F_INIT) != 0, (r[:, 5] & self.F_PRIOR) != 0
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = safety_factor * thr / dec
        nreq = np.ceil(np.maximum(np.where(np.isnan(ratio), 0.0, ratio), 0.0))
        nreq[neg | init] = 1.0
        return r, dec, thr0, neg, init, prior, nreq

    def analyse(self, rows, order, pos0, batch, max_refinement, safety_factor):
        r, dec, thr0, neg, init, prior, nreq = self._cells(rows, order, safety_factor)
        m = len(r)
        info = np.zeros((max(-(-m // batch), 1), 4), dtype=np.int32)
        first = (1 << 63) - 1
        for b, lo in enumerate(range(0, m, batch)):
            hi = min(lo + batch, m)
            unsafe = np.flatnonzero(~(prior[lo:hi] | neg[lo:hi]))
            if not len(unsafe):
                info[b] = (1, 0, 0, 0)
                continue
            bound = lo + unsafe[0]
            bad = np.flatnonzero(~((nreq[bound:hi] >= 1) & (nreq[bound:hi] <= max_refinement)))
            stop = bound + bad[0] if len(bad) else hi
            refine = bound
            if stop > bound:
                run = dec[bound:stop]
                worst = np.inf if np.isnan(run).any() else run.max()
                refined = thr0[bound:stop] * (self.tau / nreq[bound:stop])
                fail = np.flatnonzero(~(worst < refined))
                refine = bound + fail[0] if len(fail) else stop
            passes = int(stop == hi and refine == stop)
            info[b] = (passes, bound - lo, stop - bound, refine - bound)
            if not passes:
                first = min(first, pos0 // batch + b)
        return torch.from_numpy(info), torch.tensor([first], dtype=torch.int64)

    def apply(self, rows, order, info, pos0, batch, b_star, safety_factor):
        r, dec, thr0, neg, init, prior, nreq = self._cells(rows, order, safety_factor)
        m = len(r)
        safe = prior | neg
        ref = np.where(neg, 1, r[:, 4])
        for b, lo in enumerate(range(0, m, batch)):
            hi = min(lo + batch, m)
            gb = pos0 // batch + b
            passes, bound, stop, refine = (int(v) for v in info[b])
            if gb > b_star:
                safe[lo:hi], ref[lo:hi] = prior[lo:hi], r[lo:hi, 4]
                continue
            if passes and bound == 0 and stop == 0 and refine == 0:
                continue                                   # no unsafe cell in this batch
            ref[lo:hi][init[lo:hi]] = 1
            cut = lo + bound + refine
            safe[lo + bound:cut] = True
            ref[lo + bound:cut] = nreq[lo + bound:cut].astype(np.int64)
            if gb == b_star:
                safe[cut:hi], ref[cut:hi] = False, 0
        out = np.empty((m, 3), dtype=np.int64)
        slots = order.numpy()
        out[slots, 0], out[slots, 1], out[slots, 2] = r[:, 1], safe, ref
        return torch.from_numpy(out)

    def scatter(self, out_rows):
        o = out_rows.numpy()
        count = self.hi - self.lo
        self.safe = np.zeros(count, dtype=bool)
        self.refinement = np.zeros(count, dtype=np.int64)
        at = o[:, 0] - self.lo
        self.safe[at] = o[:, 1] != 0
        self.refinement[at] = o[:, 2]
        self.safe |= self.init
        self.refinement[self.init] = 1
        return torch.tensor([int(self.safe.sum())], dtype=torch.int64)


def reference_adaptive_loop(values, decrease, thr0, tau, init, prior_safe, prior_ref, batch,
                            max_refinement, safety_factor):
    """The reference's loop itself (``lyapunov.py:512-606``, adaptive branch) on per-cell arrays:
    ``decrease`` = v_decrease_bound, ``thr0`` = threshold(x, tau = 1).  -> (safe, refinement, c_max).
    TEST INFRASTRUCTURE: the yardstick for synthetic inputs the oracle's models cannot produce."""
    n = len(values)
    order = np.lexsort((np.arange(n), np_vbits(values)))           # stable (V, index) order
    safe, ref = prior_safe[order].copy(), prior_ref[order].astype(np.int64)
    threshold = thr0 * tau
    negative_all = decrease < threshold
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = safety_factor * threshold / decrease
    n_req_all = np.ceil(np.maximum(np.where(np.isnan(ratio), 0.0, ratio), 0.0))
    i = bound = refine_bound = 0
    for i in range(0, n, batch):
        idx = order[i:i + batch]
        safe_b, ref_b = safe[i:i + batch], ref[i:i + batch]                     # views
        negative = negative_all[idx]
        safe_b |= negative
        ref_b[negative] = 1
        bound, refine_bound = int(np.argmin(safe_b)), 0
        if bound > 0 or not safe_b[0]:
            n_req = n_req_all[idx[bound:]]
            ref_b[bound:] = np.where(np.isfinite(n_req) & (n_req < 2.0 ** 62), n_req, -1).astype(np.int64)
            ref_b[negative | init[idx]] = 1
            check = ((ref_b >= 1) & (ref_b <= max_refinement))[bound:]
            stop = len(check) if check.all() else int(np.argmin(check))
            if stop > 0:
                run = decrease[idx[bound:bound + stop]]
                refined = thr0[idx[bound:bound + stop]] * (tau / ref_b[bound:bound + stop])
                ok = np.array([np.all(run < r) for r in refined])
                refine_bound = len(ok) if ok.all() else int(np.argmin(ok))
                safe_b[bound:bound + refine_bound] = True
            if stop < len(check) or refine_bound < stop:
                safe_b[bound + refine_bound:] = False
                ref_b[bound + refine_bound:] = 0
                break
    c_max = values[order[i + bound + refine_bound - 1]]
    out_safe, out_ref = np.zeros(n, dtype=bool), np.zeros(n, dtype=np.int64)
    out_safe[order[safe]] = True
    out_ref[order] = ref
    out_safe[init] = True
    out_ref[init] = 1
    return out_safe, out_ref, c_max


def synthetic_adaptive_cells(n, seed, tau=0.5, hard=0.002):
    """Per-cell arrays under which WHOLE batches are accepted through refinement: the decrease is
    -1 everywhere, the thresholds ask for N in {none, 2, 3} and, rarely (``hard``), for more than any
    max_refinement of the tests; a few cells have NaN / positive decreases."""
    rng = np.random.default_rng(seed)
    values = rng.random(n).round(3)                    # ties
    decrease = np.full(n, -1.0)
    kind = rng.choice(4, n, p=[0.5, 0.3, 0.2 - hard, hard])
    thr = np.choose(kind, [-0.5, -1.5, -2.5, -40.0]) * rng.uniform(0.97, 1.0, n)
    odd = rng.random(n) < hard / 4
    decrease[odd] = rng.choice([np.nan, 0.0, 0.3], int(odd.sum()))
    init = values < 0.01
    return values, decrease, thr / tau, tau, init


def minimax_region(values, start):
    """``sl_region.hip`` in NumPy (TEST INFRASTRUCTURE): the region of ``get_lyapunov_region`` from the
    minimax-distance fixpoint D(v) = max(V(v), min over the 3^d - 1 neighbours of D), the stop level
    c* = min(D on the grid boundary, D of nodes reached only by descending), the last regular pop x*
    and the node it descends to.  ``values``: array of the grid's shape; ``start``: index tuple."""
    import itertools
    values = np.asarray(values, dtype=np.float64)
    shape, d = values.shape, values.ndim
    dist = np.full(shape, np.inf)
    dist[tuple(start)] = values[tuple(start)]
    offsets = [o for o in itertools.product((-1, 0, 1), repeat=d) if any(o)]
    pad = np.pad(dist, 1, constant_values=np.inf)
    inner = tuple(slice(1, -1) for _ in range(d))
    boundary = np.zeros(shape, dtype=bool)
    for k in range(d):
        edge = [slice(None)] * d
        for side in (0, -1):
            edge[k] = side
            boundary[tuple(edge)] = True
    while True:
        pad[inner] = dist
        best = np.full(shape, np.inf)
        for off in offsets:
            view = pad[tuple(slice(1 + o, 1 + o + n) for o, n in zip(off, shape))]
            best = np.minimum(best, view)
        cand = np.maximum(values, best)
        cand[tuple(start)] = values[tuple(start)]
        new = np.minimum(dist, cand)
        if np.array_equal(new, dist):
            break
        dist = new
    reached = np.isfinite(dist)
    stops = reached & (boundary | (dist > values))
    cstar = dist[stops].min()
    region = dist < cstar
    last = np.argwhere((dist == cstar) & (values == cstar))
    assert len(last) == 1                                     # (no ties at the stop level)
    x = tuple(last[0])
    if not boundary[x]:
        region[x] = True
        lows = [tuple(np.add(x, o)) for o in offsets]
        lows = [n for n in lows if all(0 <= a < s for a, s in zip(n, shape))
                and values[n] < cstar and dist[n] == cstar]
        if lows:
            y = min(lows, key=lambda n: values[n])
            if not boundary[y]:
                region[y] = True
    return region
