"""Grid sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The Lyapunov sweep shards by contiguous, 64-aligned ranges of the flat cell index; the only
data-path collectives are scalar reductions (the lexicographic min of the failing-cell key,
counters, the Bellman residual) and, for value iteration, an all-gather of the value table.
``torch.distributed`` is used purely as the RCCL binding; on CPU test runs the same code runs
over gloo.
"""

import os

import numpy as np

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


def _split_u64(value):
    """uint64 -> two non-negative int64 halves (RCCL / gloo min and max work on signed ints)."""
    return (value >> 32) & 0xffffffff, value & 0xffffffff


def allreduce_key(vbits, index, op, device):
    """Lexicographic min / max of ``(vbits, index)`` keys across ranks.

    One all-gather of three int64 words per rank (24 bytes): the keys are compared on the host,
    which keeps the unsigned 64-bit ordering exact (no float round trip)."""
    if not is_distributed():
        return vbits, index
    import torch
    import torch.distributed as dist
    hi, lo = _split_u64(vbits)
    mine = torch.tensor([hi, lo, index], dtype=torch.int64, device=device)
    gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    keys = [((int(g[0]) << 32) | int(g[1]), int(g[2])) for g in gathered]
    return min(keys) if op == 'min' else max(keys)


def allreduce_int(value, device):
    """SUM of one Python int across ranks."""
    if not is_distributed():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device)
    allreduce_sum_(t)
    return int(t[0])


def allreduce_sum_(tensor):
    """In-place SUM all-reduce of an integer / float tensor."""
    if is_distributed():
        import torch.distributed as dist
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def allreduce_max_(tensor):
    if is_distributed():
        import torch.distributed as dist
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
    return tensor


def allgather_concat(tensor, sizes):
    """Concatenate per-rank 1-D tensors of (possibly different) lengths ``sizes``."""
    if not is_distributed():
        return tensor
    import torch
    import torch.distributed as dist
    longest = max(sizes)
    padded = torch.zeros(longest, dtype=tensor.dtype, device=tensor.device)
    padded[:tensor.numel()] = tensor
    out = [torch.empty_like(padded) for _ in sizes]
    dist.all_gather(out, padded)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])
