"""Multi-GPU plumbing for the hot path: independent tuples shard over ranks, no data-path
collective; the single collective is the metric gather at the end (the reference's only
explicit collective is the 1-element ``all_reduce`` of the validation loss, ``train.py:102-106``).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  A few KB per evaluation: latency-bound, ring bandwidth irrelevant.
"""
import numpy as np
import torch


def shard_range(n_items, rank, world):
    """Contiguous block of item indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_pair_errors(local_errors, device=None, group=None):
    """All-gather variable-length per-pair pose errors (degrees); every rank gets the full array
    in rank order.  inf marks a pair whose pose could not be computed (``eval_pairs.py:259``)."""
    import torch.distributed as dist
    e = torch.as_tensor(np.asarray(local_errors, dtype=np.float32))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return e.numpy().astype(np.float64)
    if device is not None:
        e = e.to(device)
    world = dist.get_world_size(group)
    n = torch.tensor([e.numel()], dtype=torch.int64, device=e.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.full((m,), float("inf"), dtype=e.dtype, device=e.device)
    pad[: e.numel()] = e
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return np.concatenate([b[: int(s.item())].cpu().numpy() for b, s in zip(bufs, sizes)]).astype(np.float64)


def reduce_max_seconds(seconds, device=None, group=None):
    """MAX over ranks of a wall-clock interval (bench.py's timing rule)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
