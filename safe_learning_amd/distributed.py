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


def gather_words(tensor):
    """Every rank's copy of a small packed int64 record as ONE host array ``[world, len]``.

    This is the only communication pattern of ``update_safe_set``'s reductions: the kernels leave
    their per-shard result (failing key, last safe key, largest key, counters: 64 bytes) in
    device memory, one all-gather moves the packed records device to device (RCCL over xGMI) and
    one copy brings all of them to the host, where the lexicographic (unsigned 64-bit) comparisons
    are exact.  No per-scalar round trips."""
    import torch
    if not is_distributed():
        return tensor.detach().cpu().numpy().reshape(1, -1)
    import torch.distributed as dist
    world = dist.get_world_size()
    flat = tensor.detach().reshape(-1).contiguous()
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat)
    return out.cpu().numpy().reshape(world, -1)


def u64(word):
    """int64 bit pattern (as stored in the packed records) -> Python int in [0, 2^64)."""
    return int(word) & 0xFFFFFFFFFFFFFFFF


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
