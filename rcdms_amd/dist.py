"""Multi-GPU layer: one process per GPU, stories sharded across ranks, RCCL over xGMI only OUTSIDE the hot loop.

Reference: stage2_batchtest_rcdms_model.py:58-70,457-468 — `split_list(n_stories, n_gpus)` + one spawned process
per device, no communication at all (every process loads its own copy of every checkpoint from disk).
Frames of a story cannot be sharded (cross-frame GroupNorm + temporal attention couple them: SURVEY F2), so the
independent unit is the story.  What RCCL is used for here:
  * broadcast_module: rank 0 loads / initialises the weights once, every other rank receives them over xGMI
    (2.55 GB as f16, 5.1 GB as the fp32 state dict; bucketed so each collective is a few hundred MB);
  * broadcast_context / gather_stories: shared reference context out, finished latents back to rank 0.
The denoising loop itself contains no collective ("weak" scaling by construction) — except in the optional CFG-split
latency mode (cfg_split_layout / CfgSplit below): two GPUs per story, each evaluates one classifier-free-guidance half
of the UNet and the two 164-KB noise predictions are all-gathered inside the step graph.
Works with backend "nccl" (= RCCL on ROCm) and "gloo" (CPU tests)."""
import torch
import torch.distributed as dist


class Watchdog:
    """Bounded wait around a phase that contains collectives (weight broadcast, a CFG-split loop whose step graph holds an
    RCCL all-gather, the final gather): a peer that never arrives leaves this process blocked inside a HIP / RCCL call that
    no Python exception can interrupt, so after `seconds` a timer thread reports what was running — with the library's last
    RCCL error (rcdm_comm_last_error) — and ends the process with exit code 86; the launcher (torch.distributed.run) then
    tears the other ranks down.  A hung pair fails loudly inside the caller's own timeout instead of consuming it.
        with Watchdog(600, "rank 3: timed region"): ...
    seconds <= 0 disables it."""
    EXIT_CODE = 86

    def __init__(self, seconds, what, on_expire=None):
        self.seconds, self.what, self.on_expire, self._t = float(seconds), what, on_expire, None

    def _fire(self):
        import os
        import sys
        err = "n/a"
        try:
            from . import hip
            if hip._lib is not None:
                err = str(hip._lib.rcdm_comm_last_error())
        except Exception:
            pass
        print(f"[rcdms_amd.dist.Watchdog] '{self.what}' did not finish within {self.seconds:.0f} s — a peer rank is missing or a "
              f"collective hangs (rcdm_comm_last_error = {err}); aborting this rank", file=sys.stderr, flush=True)
        if self.on_expire is not None:
            self.on_expire()
        os._exit(self.EXIT_CODE)

    def __enter__(self):
        if self.seconds > 0:
            import threading
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()
        return self

    def __exit__(self, *exc):
        if self._t is not None:
            self._t.cancel()
        return False


def split_stories(n_stories, world_size):
    """Contiguous near-equal shards, first shards one longer (reference split_list, :58-70)."""
    base, extra = divmod(n_stories, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(list(range(start, start + n)))
        start += n
    return out


# State-dict keys whose tensors the launch planner rounds to f16 DIRECTLY (Packer.mat_f16 / conv3x3 / the chain packers) and
# composes with nothing in fp32 first: for these f16(weight) on the wire is exactly what every rank's kernels consume.  All
# other matrices are composed in fp32 before their single rounding — W diag(gamma) of the deferred LayerNorm (attn1 q|k|v,
# attn2.to_q, ff.net.0.proj, the motion modules' q|k|v), W_proj_out W_ff2 (Packer.ffz), the tap sums of the phase-form
# upsampler, W pe of the positional-encoding rows — and travel as stored, or a replica would round twice where the
# single-GPU path rounds once.
_F16_WIRE_SUFFIXES = ("conv1.weight", "conv2.weight", "conv_shortcut.weight", "downsamplers.0.conv.weight", "conv_in.weight",
                      "conv_out.weight", "to_out.0.weight", "attn2.to_k.weight", "attn2.to_v.weight", "proj_in.weight",
                      "time_emb_proj.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.weight")


def f16_wire_ok(name):
    return name.endswith(_F16_WIRE_SUFFIXES)


def broadcast_module(module, src=0, bucket_bytes=256 << 20, wire_dtype=None, wire_ok=f16_wire_ok):
    """Broadcast every parameter and buffer of `module` from rank `src`, coalesced into flat buckets so that the
    ring over point-to-point xGMI links moves a few large messages instead of 1286 small ones.  ONE flat staging
    buffer per dtype is allocated (bucket_bytes, or the largest single tensor) and reused by every bucket: `src` copies
    its tensors in, everybody copies the payload out — no per-bucket torch.cat allocation on any rank.
    wire_dtype (e.g. torch.float16): the weight matrices the kernels consume as f16(weight) and nothing else
    (`wire_ok(key)`: f16_wire_ok — the convolutions and the directly-rounded projections, ~60 % of the bytes) travel in that
    dtype; `src` rounds its own masters of exactly those keys the same way, so all replicas' state dicts and launch plans
    are bit-identical to the single-GPU path's.  Every other tensor (vectors, positional encodings, matrices that are
    composed in fp32 before their one rounding) travels as stored."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    is_src = dist.get_rank() == src
    by_dtype = {}
    for name, t in module.state_dict().items():
        if not torch.is_tensor(t):
            continue
        wire = t.dtype
        if (wire_dtype is not None and t.is_floating_point() and t.dim() >= 2 and wire_ok(name)
                and t.element_size() > torch.empty(0, dtype=wire_dtype).element_size()):
            wire = wire_dtype
        by_dtype.setdefault((wire, t.device), []).append(t)
    for (wire, device), group in by_dtype.items():
        esz = torch.empty(0, dtype=wire).element_size()
        cap = max(max(t.numel() for t in group), max(1, bucket_bytes // esz))
        flat = torch.empty(cap, dtype=wire, device=device)
        bucket, fill = [], 0

        def flush():
            nonlocal bucket, fill
            if not bucket:
                return
            view = flat[:fill]
            dist.broadcast(view, src=src)
            off = 0
            with torch.no_grad():
                for b in bucket:
                    n = b.numel()
                    b.copy_(view[off:off + n].view_as(b))     # receivers: the payload; src: its own (possibly rounded) copy
                    off += n
            bucket, fill = [], 0

        for t in group:
            n = t.numel()
            if fill + n > cap:
                flush()
            if is_src:   # (receivers have nothing worth staging: their slice of `flat` is overwritten by the broadcast)
                with torch.no_grad():
                    flat[fill:fill + n].copy_(t.detach().reshape(-1))
            bucket.append(t)
            fill += n
            if fill * esz >= bucket_bytes:
                flush()
        flush()
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


# ---------------------------------------------------------------------------------------------------------------
# CFG-split latency mode.  Reference arithmetic: RCDMs_pipeline.py:482-497 — latent_model_input = cat([latents] * 2),
# noise_pred_uncond, noise_pred_text = noise_pred.chunk(2); batch elements never interact inside the UNet, so the two
# halves can run on two devices and only the noise predictions meet.

def cfg_split_layout(world_size, rank):
    """(pair index, half, partner rank) of `rank` when ranks are paired (0,1), (2,3), ...: the even rank of a pair
    evaluates the unconditional half (batch rows [0, S)), the odd rank the conditional half ([S, 2S)) — the order of
    the reference's CFG batch.  Story shards are dealt to PAIRS: split_stories(n, world_size // 2)[pair]."""
    if world_size < 2 or world_size % 2:
        raise ValueError(f"CFG split pairs the ranks: world size {world_size} is not a positive even number")
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return rank // 2, rank % 2, rank ^ 1


def exchange_unique_id(make_id, world_size, rank):
    """Every pair's even rank draws an id with make_id() (rcdms_amd.hip.Comm.unique_id); returns the id of this
    rank's pair on both of its members.  One all_gather_object over the default group (any backend)."""
    mine = make_id() if rank % 2 == 0 else None
    ids = [None] * world_size
    dist.all_gather_object(ids, mine)
    return ids[rank - (rank % 2)]


class CfgSplit:
    """What DenoiseLoop needs to run one CFG half: which half, and `allgather(send_ptr, recv_ptr, nbytes)` enqueueing
    the exchange of the two halves' noise predictions on the current stream (recv holds [uncond | cond])."""

    def __init__(self, half, allgather):
        self.half = int(half)
        self.allgather = allgather

    @classmethod
    def from_world(cls):
        """Pair communicators over RCCL through the C-ABI (rcdm_comm_*), ids exchanged through torch.distributed."""
        from . import hip
        world, rank = dist.get_world_size(), dist.get_rank()
        _, half, _ = cfg_split_layout(world, rank)
        uid = exchange_unique_id(hip.Comm.unique_id, world, rank)
        comm = hip.Comm(uid, 2, half)
        obj = cls(half, comm.allgather)
        obj.comm = comm
        return obj

    def close(self):
        """Destroy the pair communicator (both ranks call it, after the DenoiseLoop that captured the exchange is dropped)."""
        comm = getattr(self, "comm", None)
        if comm is not None:
            comm.close()
            self.comm = None


def cfg_split_reference_step(eps_fn, x, ctx_u, ctx_c, half, allgather_tensors):
    """Host restatement of one split step's data flow (CPU tests over gloo): this rank evaluates eps_fn on ITS half
    only, the halves are all-gathered, and the caller applies CFG + DDIM to the gathered pair exactly as the unsplit
    loop does.  Returns (eps_uncond, eps_cond)."""
    mine = eps_fn(x, ctx_u if half == 0 else ctx_c).contiguous()
    both = [torch.empty_like(mine), torch.empty_like(mine)]
    allgather_tensors(both, mine)
    return both[0], both[1]
