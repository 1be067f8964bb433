"""Multi-GPU sharding of tracked points (the reference's `parallel=` of tracker.py:381-387).

Tracks are independent (each owns its particles, weights, templates and random draws;
SURVEY.md 8(e)), so the points are split into contiguous blocks, one block per process =
per GPU, and nothing is exchanged while a sequence runs.  At the end the per-point
posterior moments are collected on rank 0 with ONE gather (RCCL over xGMI for device
tensors, gloo for host tensors), in `motion_models` order.

    torchrun --nproc-per-node 8 script.py      # one rank per GPU
    rank, world = sharding.init()              # torch.distributed, device = LOCAL_RANK
    lo, hi = sharding.shard_range(len(models), world, rank)
    tracks = tracker.track(models[lo:hi], ..., point_offset=lo)
    means, sigmas = sharding.gather_points([tracks.means, tracks.sigmas], len(models))
"""
import os

import numpy as np


def shard_range(n_points, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the remainder goes to the first ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    base, extra = divmod(int(n_points), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_points, world):
    return [hi - lo for lo, hi in (shard_range(n_points, world, r) for r in range(world))]


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun) and bind
    this process to GPU LOCAL_RANK.  Returns (rank, world); (0, 1) without a launcher."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


def gather_points(arrays, n_points, dst=0, group=None):
    """Gather per-point arrays (leading axis = this rank's points) to rank `dst`.

    `arrays`: list of ndarrays (host) or torch tensors (host or device) whose first axis has
    this rank's `shard_range` length.  Returns the list of full arrays (first axis
    `n_points`, in global point order) on `dst`, None elsewhere.  One collective per array;
    shards are padded to the largest shard so a plain `gather` is enough (no all-to-all)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(arrays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_points, world)
    biggest = max(sizes)
    out = []
    for a in arrays:
        t = torch.as_tensor(a) if not isinstance(a, torch.Tensor) else a
        if t.shape[0] != sizes[rank]:
            raise ValueError(f"rank {rank}: expected {sizes[rank]} points, got {t.shape[0]}")
        if t.shape[0] < biggest:
            pad = torch.zeros((biggest - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad])
        t = t.contiguous()
        recv = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, recv, dst=dst, group=group)
        if rank == dst:
            full = torch.cat([recv[r][: sizes[r]] for r in range(world)])
            out.append(full.cpu().numpy() if isinstance(a, np.ndarray) else full)
    return out if rank == dst else None
