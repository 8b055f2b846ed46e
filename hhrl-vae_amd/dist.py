"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests).

The learning path shards by minibatch rows and has ONE exchange step per
optimizer step: a SUM all-reduce of the flat gradient buffer.  Loss terms that
are means over the batch are divided by the GLOBAL count on every shard, terms
that are sums are left alone, so that the summed shard gradients equal the
single-device gradient of the global batch and every rank applies the same
Adam update to its replica.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def world(group) -> Tuple[int, int]:
    """(rank, world_size) of `group`; (0, 1) without a group."""
    if group is None:
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(group), dist.get_world_size(group)


def shard_bounds(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Rows [lo, hi) of a global batch owned by `rank`: contiguous, sizes
    differ by at most one, every row owned exactly once."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_host_staged = {}


def host_staged(group) -> bool:
    """True for a gloo group.  gloo is the backend of CPU tests and of rehearsals of the N-rank path with
    several ranks on one GPU (RCCL is the production backend).  Its own handling of device tensors (side
    streams, events, pinned staging buffers) was seen to stall when several processes share one GPU, so
    device tensors never reach it: the helpers below copy them to the host, reduce there, and copy back.
    RCCL collectives take the device tensor and queue behind the kernels on the launch stream."""
    key = id(group)
    v = _host_staged.get(key)
    if v is None:
        import os
        import torch.distributed as dist
        # HHRL_GLOO_DEVICE_TENSORS=1 (diagnostic, tools/dp_hang_probe.py): hand gloo the device tensor itself
        v = str(dist.get_backend(group)) == "gloo" and os.environ.get("HHRL_GLOO_DEVICE_TENSORS") != "1"
        _host_staged[key] = v
    return v


def global_count(local_count: int, group, device=None) -> int:
    """Sum of `local_count` over the group (one tiny all-reduce + host read)."""
    return int(round(reduce_scalar(float(local_count), "sum", group, device)))


def reduce_scalar(value: float, op: str, group, device=None) -> float:
    """`value` reduced over the group with op in {"sum", "min", "max"} and read back on the host: the
    control-plane reductions of a data-parallel run (element counts, agreed step counts, the slowest rank's
    time).  One tiny all-reduce + a host read; the tensor lives on `device` (RCCL reduces device tensors only)."""
    if group is None:
        return float(value)
    import torch.distributed as dist
    ops = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if host_staged(group) else device)
    dist.all_reduce(t, op=ops[op], group=group)
    return float(t.item())


def allreduce_sum_(flat: torch.Tensor, group) -> torch.Tensor:
    """In-place SUM all-reduce of one flat fp32 bucket (no-op without a group)."""
    if group is not None:
        import torch.distributed as dist
        if flat.is_cuda and host_staged(group):
            h = flat.detach().cpu()                   # waits for the launch stream
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_ints(values, group, device=None):
    """The group's first rank's `values` (a short sequence of ints) on every rank: decisions that the
    ranks draw from their own host generators but must take alike, such as the order in which a
    multi-actor update steps its actors (algorithm.py:809) — each actor's step has an all-reduce of its
    own, so the ranks have to walk the actors in the same order.  No-op without a group."""
    vals = [int(v) for v in values]
    if group is None:
        return vals
    import torch.distributed as dist
    t = torch.tensor(vals, dtype=torch.int64, device=None if host_staged(group) else device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
    return [int(v) for v in t.cpu().tolist()]
