"""Frame sharding across the GPUs of one node.

Frames are independent decodes: contiguous frame ranges per rank, H replicated,
NO collective on the decode path.  The only exchange is the final gather of decoded blocks (packed
hard decisions + per-frame status) to rank 0 -- RCCL over xGMI on GPUs (backend "nccl"), gloo on CPU.
"""
import torch
import torch.distributed as dist


def frame_range(total_frames, world_size, rank):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra frame."""
    base, extra = divmod(int(total_frames), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_blocks(local, total_frames, dst=0, group=None):
    """Gather per-frame rows (dim 0 = this rank's frame_range) to `dst` in global frame order.

    Returns the [total_frames, ...] tensor on dst and None elsewhere.  Shards may be ragged, so every
    rank pads to the largest shard and dst trims; one all_gather-free `dist.gather` is the whole exchange.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [frame_range(total_frames, world, r) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    assert local.shape[0] == hi - lo, "local rows do not match this rank's frame range"
    if local.shape[0] < cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) != "nccl":
        local = local.cpu()            # gloo rehearsal: stage through host memory
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[: h - l] for b, (l, h) in zip(bufs, sizes)], 0)
