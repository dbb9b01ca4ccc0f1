"""Clip-level data parallelism: the only parallelism the path has.

Clips are independent units (eval BatchNorm, no cross-sample op), so a global batch is cut into contiguous
per-rank shards, every rank runs the full model on its shard, and the only collectives are a one-off weight
broadcast and a per-step map gather.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n units for `rank` (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(module, src=0):
    """One flat broadcast of every floating-point parameter and buffer (bucketed into a single message:
    xGMI links are point-to-point, so one large transfer beats hundreds of small ones)."""
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    if hasattr(module, "_invalidate"):
        module._invalidate()        # packed weights are stale now
    return flat.numel()


def gather_maps(local_maps, n_total, dst=0):
    """Gather per-rank [n_r, H, W] maps to `dst` in global clip order (ragged shards allowed)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = local_maps.new_zeros((nmax,) + tuple(local_maps.shape[1:]))
    pad[: local_maps.shape[0]] = local_maps
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
