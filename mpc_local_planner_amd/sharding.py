"""Multi-GPU sharding of a batch of OCP instances.

Instances are independent, so the batch is cut into contiguous blocks, one per rank, with no
collective on the data path.  The only exchange is one all-gather of the fixed-size control
records (u0[2], x1.delta, x1.throttle, cost, status+16*iters -- include/nmpc.h:
nmpc_get_controls) at the end of a step; on GPUs it runs over RCCL/xGMI straight from the
handle's device buffer, on CPU (tests) over gloo.
"""
import numpy as np

RECORD = 6   # doubles per control record


def shard_range(rank, world, total):
    """Contiguous block [start, stop) of `total` instances owned by `rank` (blocks differ by at
    most one instance, lower ranks take the remainder)."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank/world/total")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(world, total):
    return [shard_range(r, world, total)[1] - shard_range(r, world, total)[0] for r in range(world)]


def gather_controls(ctrl_local, total=None, group=None):
    """All-gather the [B_local, 6] control records of every rank into [total, 6] on every rank.
    `ctrl_local` is a torch tensor (CPU with gloo, CUDA with nccl = RCCL).  Equal shard sizes use
    one all_gather_into_tensor; ragged shards are padded to the largest one."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = ctrl_local.shape[0]
    if total is None:
        total = n_local * world
    sizes = shard_sizes(world, total)
    if sizes[rank] != n_local:
        raise ValueError(f"rank {rank} holds {n_local} records, expected {sizes[rank]}")
    nmax = max(sizes)
    flat = ctrl_local.reshape(-1)
    if n_local < nmax:
        flat = torch.cat([flat, flat.new_zeros((nmax - n_local) * RECORD)])
    out = flat.new_empty(world * nmax * RECORD)
    dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
    out = out.reshape(world, nmax, RECORD)
    if min(sizes) == nmax:
        return out.reshape(total, RECORD)
    return torch.cat([out[r, :sizes[r]] for r in range(world)], dim=0)
