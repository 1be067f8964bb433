"""For callers inside a `torch.distributed` job (gloo on CPU, nccl = RCCL on GPUs): the end-of-sequence gather of
glimpse_amd issued by the CALLER instead of by the library.  An example, not part of the product: `glimpse_amd` itself
never imports torch -- its own gather is RCCL behind the C ABI (`glh_gather_moments`, glimpse_amd/sharding.py: Group).

    rank, world = torch_interop.init()                      # joins the process group of torchrun's environment
    lo, hi = sharding.shard_range(len(models), world, rank)
    ... track the block on this rank's GPU ...
    ptr, nbytes = ctx.moments_device()                      # the history stays on the device
    mine = torch.as_tensor(torch_interop.DeviceArray(ptr, (T, hi - lo, 12)), device=f"cuda:{local_rank}")   # zero copy
    full = torch_interop.gather_points([mine], len(models))  # rank 0: (sum P, ...) in track order

Import torch BEFORE glimpse_amd loads libglimpse_hip.so (torch bundles its own HIP runtime).
"""
import os

import numpy as np

from glimpse_amd.sharding import shard_sizes


class DeviceArray:
    """Zero-copy view of a library-owned device buffer through `__cuda_array_interface__` (torch.as_tensor accepts
    it): e.g. the moments history of `Context.moments_device()` for a collective the caller issues itself."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


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
    """Gather per-point arrays (leading axis = this rank's points) to rank `dst` with torch.distributed.

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
