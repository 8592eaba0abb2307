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


def gather_buffer(local, total_frames, dst=0, group=None):
    """Receive buffer for gather_blocks on `dst`: [world, cap, ...] (cap = the largest shard), allocated ONCE by the caller so that
    a timed step holds the collective and nothing else.  None on the other ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or dist.get_rank(group) != dst:
        return None
    world = dist.get_world_size(group)
    cap = max(hi - lo for lo, hi in (frame_range(total_frames, world, r) for r in range(world)))
    dev = local.device if dist.get_backend(group) == "nccl" or not local.is_cuda else torch.device("cpu")
    return torch.empty((world, cap) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)


def gather_blocks(local, total_frames, dst=0, group=None, out=None):
    """Gather per-frame rows (dim 0 = this rank's frame_range) to `dst` in global frame order.

    Returns the [total_frames, ...] tensor on dst and None elsewhere: ONE `dist.gather` is the whole exchange (RCCL over xGMI on GPUs).
    With `out` = gather_buffer(...) the rows land in that preallocated buffer and, when every rank holds the same number of
    frames, the result is a view of it (no allocation, no copy); ragged shards are padded to the largest one and trimmed on dst.
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
    bufs = None
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        assert tuple(out.shape) == (world,) + tuple(local.shape) and out.dtype == local.dtype and out.device == local.device
        bufs = [out[r] for r in range(world)]
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if all(h - l == cap for l, h in sizes):
        return out.view((world * cap,) + tuple(local.shape[1:]))
    return torch.cat([out[r, : h - l] for r, (l, h) in enumerate(sizes)], 0)
