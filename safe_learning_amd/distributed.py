"""Grid sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The Lyapunov sweep shards by contiguous, 64-aligned ranges of the flat cell index; the only
data-path collectives are scalar reductions (the lexicographic min of the failing-cell key,
counters, the Bellman residual) and, for value iteration, an all-gather of the value table.
``torch.distributed`` is used purely as the RCCL binding; on CPU test runs the same code runs
over gloo.
"""

import os

_MASK63 = (1 << 63) - 1


def is_distributed():
    """True when the collectives have to run: more than one rank, or SL_FORCE_COLLECTIVES=1 with an
    initialised process group (lets a single-GPU box exercise the RCCL calls at world size 1)."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("SL_FORCE_COLLECTIVES") == "1"
    except Exception:
        return False


def rank_and_world():
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(), dist.get_world_size()


def shard_bounds(n, world, align=64):
    """``world + 1`` boundaries of contiguous shards, every interior one a multiple of ``align``."""
    per = -(-n // world)
    per = -(-per // align) * align
    return [min(r * per, n) for r in range(world)] + [n]


def shard_range(n, rank=None, world=None, align=64):
    r, w = rank_and_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    bounds = shard_bounds(n, world, align)
    return bounds[rank], bounds[rank + 1]


# ---- collective timing (bench.py: ``config.collective_ms``) ---------------------------------
_timing = None          # None, or a list of (start, stop) events / float seconds per collective


def start_timing():
    """Record the duration of every collective issued through this module from now on."""
    global _timing
    _timing = []


def stop_timing():
    """-> milliseconds spent inside collectives since ``start_timing`` (device events on the
    stream the collective was issued on for GPU tensors - the span includes waiting for the
    slowest rank to arrive -, host clock for CPU tensors).  Synchronises the device."""
    global _timing
    spans, _timing = _timing or [], None
    total = 0.0
    for span in spans:
        if isinstance(span, tuple):
            span[1].synchronize()
            total += span[0].elapsed_time(span[1])
        else:
            total += 1e3 * span
    return total


class _timed(object):
    """Context manager around one collective on ``tensor``'s device."""

    def __init__(self, tensor):
        self.cuda = _timing is not None and getattr(tensor, 'is_cuda', False)
        self.host = _timing is not None and not self.cuda

    def __enter__(self):
        if self.cuda:
            import torch
            self.start = torch.cuda.Event(enable_timing=True)
            self.stop = torch.cuda.Event(enable_timing=True)
            self.start.record()
        elif self.host:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.cuda:
            self.stop.record()
            _timing.append((self.start, self.stop))
        elif self.host:
            import time
            _timing.append(time.perf_counter() - self.t0)
        return False


def gather_records(tensor):
    """Every rank's packed 64-byte record in DEVICE memory -> ``(records, world)``: one
    ``all_gather_into_tensor`` (RCCL over xGMI), nothing copied to the host - the fold and the
    kernels that consume the folded record run on the device (``sl_fold_results``)."""
    if not is_distributed():
        return tensor, 1
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    flat = tensor.detach().reshape(-1).contiguous()
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    with _timed(flat):
        dist.all_gather_into_tensor(out, flat)
    return out, world


def u64(word):
    """int64 bit pattern (as stored in the packed records) -> Python int in [0, 2^64)."""
    return int(word) & 0xFFFFFFFFFFFFFFFF


def allreduce_sum_(tensor):
    """In-place SUM all-reduce of an integer / float tensor."""
    if is_distributed():
        import torch.distributed as dist
        with _timed(tensor):
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def allreduce_min_(tensor):
    if is_distributed():
        import torch.distributed as dist
        with _timed(tensor):
            dist.all_reduce(tensor, op=dist.ReduceOp.MIN)
    return tensor


def exchange_rows(rows, send_counts, recv_counts=None):
    """All-to-all of the rows of a 2-D tensor: the first ``send_counts[0]`` rows go to rank 0, the
    next ``send_counts[1]`` to rank 1, ...; returns ``(received rows, rows received per rank)`` in
    source-rank order.  ``recv_counts``: known from an earlier exchange in the other direction,
    otherwise the counts are exchanged first (one small all-to-all)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    assert len(send_counts) == world and sum(send_counts) == len(rows)
    if recv_counts is None:
        theirs = torch.empty(world, dtype=torch.int64, device=rows.device)
        mine = torch.tensor(send_counts, dtype=torch.int64, device=rows.device)
        with _timed(mine):
            dist.all_to_all_single(theirs, mine)
        recv_counts = [int(v) for v in theirs.cpu()]
    out = torch.empty((sum(recv_counts),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    with _timed(rows):
        dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=recv_counts,
                               input_split_sizes=list(send_counts))
    return out, recv_counts


def allreduce_max_(tensor):
    if is_distributed():
        import torch.distributed as dist
        with _timed(tensor):
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
    return tensor


def allgather_equal(buffer, total=None, out=None):
    """ONE ``all_gather_into_tensor`` of equally sized per-rank buffers into a pre-sized tensor
    (``out`` is reused when given); returns the first ``total`` elements.

    Contiguous shards (``shard_bounds``) are all ``per`` elements long except the last non-empty
    one, so gathering the shards padded to ``per`` and cutting the result at ``total`` IS the
    concatenation: no per-rank list, no ``torch.cat``, no extra copy of the gathered data."""
    import torch
    flat = buffer.reshape(-1)
    if not is_distributed():
        return flat if total is None else flat[:total]
    import torch.distributed as dist
    world = dist.get_world_size()
    if out is None or out.numel() != world * flat.numel() or out.dtype != flat.dtype \
            or out.device != flat.device:
        out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    with _timed(flat):
        dist.all_gather_into_tensor(out, flat.contiguous())
    return out if total is None else out[:total]


def allgather_concat(tensor, sizes, out=None):
    """Concatenate per-rank 1-D tensors of lengths ``sizes`` (the same list on every rank).

    Sizes of contiguous shards (``longest`` on the leading ranks, one shorter shard, then empty
    ones) take one ``all_gather_into_tensor`` and a slice; any other pattern is compacted after
    the same single collective."""
    if not is_distributed():
        return tensor
    import torch
    longest = max(sizes)
    if longest == 0:
        return tensor
    if tensor.numel() == longest:
        padded = tensor.reshape(-1)
    else:
        padded = torch.zeros(longest, dtype=tensor.dtype, device=tensor.device)
        padded[:tensor.numel()] = tensor.reshape(-1)
    full = allgather_equal(padded, out=out)
    short = [r for r, s in enumerate(sizes) if s < longest]
    if all(s == 0 for r, s in enumerate(sizes) if short and r > short[0]):
        return full[:sum(sizes)]
    return torch.cat([full[r * longest:r * longest + s] for r, s in enumerate(sizes)])
