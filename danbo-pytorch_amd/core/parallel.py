"""Multi-GPU plumbing: one process per GPU, rays sharded data-parallel.

Rendering has no data-path collective: every rank renders a contiguous block of rays (or its
own camera views) with the full model and recomputes the 1.7 MMAC pose GNN; the only
communication is the optional gather of finished per-ray maps (20 B/ray) to rank 0.
The helpers below are backend-agnostic (`nccl` = RCCL on the GPUs, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced split of n items: ranks < n % world get one extra item."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_rays(ray_batch, rank=None, world=None, **per_ray):
    """Slice the ray batch (and any per-ray tensors) for this rank."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    a, b = shard_range(ray_batch.shape[0], rank, world)
    return ray_batch[a:b], {k: (v[a:b] if torch.is_tensor(v) and v.shape[0] == ray_batch.shape[0] else v)
                            for k, v in per_ray.items()}


def gather_maps(local, n_total, dst=0, group=None):
    """Assemble per-ray maps (dict of [r_local, ...] tensors) on rank `dst`, in ray order.
    Uneven shards are padded to the largest shard for the all_gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    pad = max(b - a for a, b in sizes)
    out = {}
    for k, v in local.items():
        buf = torch.zeros((pad,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        buf[: v.shape[0]] = v
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        if rank == dst:
            out[k] = torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], 0)
    return out if rank == dst else None


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (bench.py's timing rule)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
