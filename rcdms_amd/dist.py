"""Multi-GPU layer: one process per GPU, stories sharded across ranks, RCCL over xGMI only OUTSIDE the hot loop.

Reference: stage2_batchtest_rcdms_model.py:58-70,457-468 — `split_list(n_stories, n_gpus)` + one spawned process
per device, no communication at all (every process loads its own copy of every checkpoint from disk).
Frames of a story cannot be sharded (cross-frame GroupNorm + temporal attention couple them: SURVEY F2), so the
independent unit is the story.  What RCCL is used for here:
  * broadcast_module: rank 0 loads / initialises the weights once, every other rank receives them over xGMI
    (2.55 GB as f16, 5.1 GB as the fp32 state dict; bucketed so each collective is a few hundred MB);
  * broadcast_context / gather_stories: shared reference context out, finished latents back to rank 0.
The denoising loop itself contains no collective ("weak" scaling by construction).
Works with backend "nccl" (= RCCL on ROCm) and "gloo" (CPU tests)."""
import torch
import torch.distributed as dist


def split_stories(n_stories, world_size):
    """Contiguous near-equal shards, first shards one longer (reference split_list, :58-70)."""
    base, extra = divmod(n_stories, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(list(range(start, start + n)))
        start += n
    return out


def broadcast_module(module, src=0, bucket_bytes=256 << 20):
    """Broadcast every parameter and buffer of `module` from rank `src`, coalesced into flat buckets so that the
    ring over point-to-point xGMI links moves a few large messages instead of 1286 small ones."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    tensors = [t for t in module.state_dict().values() if torch.is_tensor(t)]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, group in by_dtype.items():
        bucket, size = [], 0
        for t in group + [None]:
            if t is not None:
                bucket.append(t)
                size += t.numel() * t.element_size()
            if bucket and (t is None or size >= bucket_bytes):
                flat = torch.cat([b.detach().reshape(-1) for b in bucket])
                dist.broadcast(flat, src=src)
                off = 0
                for b in bucket:
                    n = b.numel()
                    with torch.no_grad():
                        b.copy_(flat[off:off + n].view_as(b))
                    off += n
                bucket, size = [], 0
    if hasattr(module, "_programs"):
        module._programs = {}


def broadcast_context(ctx, src=0):
    """Shared text/reference context (R*S*f, L, D) from rank `src` to all ranks."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(ctx, src=src)
    return ctx


def gather_stories(latents, dst=0):
    """Collect every rank's finished latents (S_r,4,f,h,w) on rank `dst`, concatenated in rank order.  Shards may be
    uneven (split_stories hands the first n % world ranks one story more): shard sizes are exchanged first and every
    rank pads to the longest, so the one data collective always sees equal shapes; `dst` trims the padding.  On RCCL
    this is a true gather (ncclGather-style send/recv to one root), not an all_gather to every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return latents
    world, rank = dist.get_world_size(), dist.get_rank()
    latents = latents.contiguous()
    n = torch.tensor([latents.shape[0]], dtype=torch.int64, device=latents.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if latents.shape[0] < longest:
        pad = torch.zeros((longest - latents.shape[0],) + tuple(latents.shape[1:]), dtype=latents.dtype,
                          device=latents.device)
        latents = torch.cat([latents, pad])
    bufs = [torch.empty_like(latents) for _ in range(world)] if rank == dst else None
    dist.gather(latents, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])
