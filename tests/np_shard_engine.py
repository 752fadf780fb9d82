"""NumPy stand-in for the HIP shard engine (TEST INFRASTRUCTURE).

Implements the three primitives that ``safe_learning_amd.lyapunov.prefix_rule`` needs with the
same semantics as the kernels (sl_lyap_sweep / sl_lyap_finalize / sl_select_pass), on a shard of
the grid, from the oracle's per-cell ``negative`` mask.  It lets the multi-rank orchestration
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

    def finalize(self, star, keep, use_prior):
        below = _lex_less(self.vb, self.idx, star)
        self.safe = self.init | below
        if use_prior:
            self.safe |= self.prior & ~_lex_less(self.vb, self.idx, keep)
        return self._record(last=self._extreme(below, largest=True),
                            largest=self._extreme(np.ones_like(below), largest=True),
                            below=int(below.sum()))

    def select_hist(self, which, byte, prefix, vbits_equal):
        key = self.vb if which == 0 else self.idx.astype(np.uint64)
        take = np.ones(len(key), dtype=bool) if which == 0 else (self.vb == np.uint64(vbits_equal))
        shift = np.uint64(8 * byte)
        if byte < 7:
            himask = np.uint64((0xFFFFFFFFFFFFFFFF << (8 * byte + 8)) & 0xFFFFFFFFFFFFFFFF)
            take &= (key & himask) == (np.uint64(prefix) & himask)
        digits = ((key[take] >> shift) & np.uint64(0xFF)).astype(np.int64)
        return torch.from_numpy(np.bincount(digits, minlength=256).astype(np.int64))
