#!/usr/bin/env python
"""bench.py — story-frames/sec of the stage-2 denoising hot path on MI355X.

Workload (BASELINE.json configs[1]): PororoSV stage-2, 512x512 (64x64 latents), 50-step DDIM, CFG (guidance 2.0),
batch = 1 story x 5 frames per GPU, context 85 x 768, synthetic inputs + random-init weights of the real
1276.9 M-parameter architecture (no checkpoints/datasets exist here).  A "step" is ONE pass of the hot path over
one batch: the full T-step denoising of `--stories` stories = T x [UNet (b = 2 S) + CFG + DDIM], inputs resident
in HBM.  N > 1: one process per GPU (torch.distributed / RCCL), stories sharded one batch per rank, NO collective
in the data path (stories are independent: SURVEY §8e) -> weak scaling; value = all ranks' frames / max-rank time.

Prints ONE JSON line (rank 0).  `roofline`: bound = mfma; one "launch" = one replay of the captured step graph
(~600 kernels = one UNet call + CFG/DDIM); achieved = 11.044 TFLOP algorithmic (SURVEY §8d, 2*MAC of the
reference's conv/addmm/mm/bmm/baddbmm at b=2,f=5,64x64,L=85) x S / the average replay duration measured with HIP
events on the launch stream.  The two CFG halves of a step have identical inputs up to the first cross-attention
(RCDMs_pipeline.py:481-482), so conv_in, the first ResNet block and the first self-attention are evaluated once and
stored for both (0.21 of the 11.044 TFLOP; exact — config.shared_cfg_prefix, --no-share-prefix for the A/B); `achieved`
still prices the reference's full 11.044 TFLOP per call (`roofline.flops_skipped_tflop_per_launch` and `achieved_issued` give the issued-work view;
the breakdown: the shared prefix; the 5/9 (four-phase form) or 27/36 (nine tap planes over the source pixels, round 6) of the Upsample3D convs' products that their exact forms do not multiply; the 20/36 the Winograd F(2x2, 3x3) form of the deep ResNet convolutions does not multiply).  `cpu_baseline`: the oracle restatement of the reference's CPU path (kind "port")
timed on this box's host cores on a bounded sample (a few UNet calls of the 50), extrapolated to T calls."""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_TFLOP_PER_CALL = {64: 11.044, 32: 2.556}  # per UNet call at b=2 (one story with CFG), L=85  [SURVEY §8d]
SHARED_PREFIX_TFLOP = {64: 0.208, 32: 0.032}   # per story and call NOT issued under the shared CFG prefix (half of 2 convs 320->320, the first self-attention, 4 C x C / qkv GEMMs)
MFMA_F16_PEAK_TFLOPS = 2500.0                  # MI355X dense fp16, MI355X_MICROARCH.md


def upsample_phase_tflop(plan):
    """TFLOP per call the plan's phase-form upsampling convs leave out against nearest-2x + conv3x3 (5 of 9 taps)."""
    import re
    t = 0.0
    for tag in plan.tags:
        m = re.match(r"conv3x3 (\d+)x(\d+)x(\d+) (\d+)->(\d+) s=1 up=2", tag)
        if m:
            n, h, w, ci, co = (int(v) for v in m.groups())
            t += 2.0 * n * 4 * h * w * 5 * ci * co / 1e12
    return t


def upsample_taps_tflop(plan):
    """TFLOP per call the tap-plane upsampling convs (one N = 9 c GEMM over the source pixels + gather) leave out against
    nearest-2x + conv3x3: 27 of the 36 products per source pixel."""
    import re
    t = 0.0
    for tag in plan.tags:
        m = re.match(r"upsample_gather (\d+)x(\d+)x(\d+) C=(\d+)", tag)
        if m:
            n, h, w, c = (int(v) for v in m.groups())
            t += 2.0 * n * h * w * 27 * c * c / 1e12
    return t


def winograd_tflop(plan):
    """TFLOP per call the Winograd F(2x2, 3x3) convs leave out against the nine taps: 20 of every 36 multiply-adds of the 3x3
    part (the 1x1 shortcut entries are plain GEMMs: nothing left out)."""
    import re
    t = 0.0
    for tag in plan.tags:
        m = re.match(r"conv3x3_wino (\d+)x(\d+)x(\d+) (\d+)->(\d+)", tag)
        if m:
            n, h, w, ci, co = (int(v) for v in m.groups())
            t += 2.0 * n * h * w * 9 * ci * co * (20.0 / 36.0) / 1e12
    return t


def init_weights_(model, seed=0):
    """Random init ON the GPU of the real architecture: unit-gain N(0, 1/fan_in) matrices, norms ~ 1, small biases."""
    from rcdms_amd.synth import _is_norm_param, sinusoid_table
    g = torch.Generator(device=model.device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.state_dict().items():
            if name.endswith("pos_encoder.pe"):
                p.copy_(sinusoid_table(p.shape[2], p.shape[1]))
            elif p.dim() == 1:
                is_norm_w = name.endswith(".weight") and _is_norm_param(name)
                p.normal_(0, 1, generator=g)
                p.mul_(0.1 if is_norm_w else 0.05)
                if is_norm_w:
                    p.add_(1.0)
            else:
                fan_in = p[0].numel()
                p.normal_(0, fan_in ** -0.5, generator=g)


def build_model(device, width=0):
    """The stage-2 UNet (SD-1.5 widths 320 / 640 / 1280 / 1280, 1276.9 M parameters); width > 0: the same topology at that
    base width with a width-wide context (harness tests only)."""
    from src.models.unet import UNet3DConditionModel
    mk = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
              temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)
    extra = dict(block_out_channels=(width, 2 * width, 4 * width, 4 * width), cross_attention_dim=width) if width else \
        dict(cross_attention_dim=768)
    with torch.device("meta"):
        m = UNet3DConditionModel(sample_size=64, in_channels=9, use_motion_module=True,
                                 motion_module_resolutions=[1, 2, 4, 8], unet_use_cross_frame_attention=False,
                                 unet_use_temporal_attention=False, motion_module_type="Vanilla",
                                 motion_module_kwargs=mk, **extra)
    m = m.to_empty(device=device).eval()
    init_weights_(m)
    return m


def cpu_baseline(model, latent, ctx_len, ddim_steps, budget_s=200.0):
    """Oracle (CPU fp32 restatement of the reference, kind "port") on this box's host cores, per BASELINE.md section 3:
    one warm-up UNet call at the REAL size (b=2, f=5, latent x latent, L=ctx_len), then up to 3 timed calls of the same;
    the median is reported and extrapolated x ddim_steps (a full 50-step CPU story is most of an hour).  Bounded: when the
    warm-up shows that three more calls do not fit budget_s, fewer are timed (at least one) and `sample` says how many.
    torch CPU scales badly past a few dozen threads, so at most 64 are used; `cores` reports the threads used."""
    from oracle import unet_oracle as O
    from rcdms_amd import synth
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    st = synth.synthetic_story(stories=1, latent_hw=(latent, latent), ctx_len=ctx_len, seed=42)
    x = torch.cat([torch.cat([st["latents"]] * 2), st["mask"], st["masked_latents"]], dim=1)

    def one_call():
        t0 = time.time()
        with torch.no_grad():
            O.unet_forward(sd, O.SD15_STAGE2_CONFIG, x, torch.tensor(981), st["ctx"])
        return time.time() - t0

    warm = one_call()
    n = max(1, min(3, int((budget_s - warm) / max(warm, 1e-3))))
    ts = sorted(one_call() for _ in range(n))
    t_call = ts[len(ts) // 2] if n % 2 else 0.5 * (ts[n // 2 - 1] + ts[n // 2])
    how = (f"1 warm-up ({warm:.1f} s) + {n} timed UNet calls of the {ddim_steps} per story at {latent}x{latent} latents: "
           f"median {t_call:.1f} s, min {ts[0]:.1f}, max {ts[-1]:.1f}")
    fps = 5.0 / (ddim_steps * t_call)
    return {"value": fps, "unit": "story-frames/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": how + f"; b=2 f=5 L={ctx_len}, fp32 torch CPU restatement on {threads} threads of this box's "
                            f"{os.cpu_count()} host cores, extrapolated x{ddim_steps} steps"}


def ddim():
    from rcdms_amd.scheduler import DDIMScheduler
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)


def time_stage2(model, S, latent, L, T, passes, structure="dense", guidance=2.0):
    """One secondary stage-2 configuration on the SAME model: 1 warm-up pass + `passes` timed passes of the T-step loop
    (HIP events on the loop's stream around every pass + the wall clock around all of them)."""
    from rcdms_amd import hip, synth
    from rcdms_amd.sampler import DenoiseLoop
    loop = DenoiseLoop(model, S, 5, latent, latent, L, guidance, ddim(), T)
    st = synth.synthetic_story(stories=S, latent_hw=(latent, latent), ctx_len=L, seed=42, structure=structure)

    def one():
        loop.load(st["latents"], st["mask"], st["masked_latents"], st["ctx"])
        loop.run()
    one()
    torch.cuda.synchronize()
    ev0, ev1 = hip.Event(), hip.Event()
    gpu_ms, t0 = 0.0, time.perf_counter()
    for _ in range(passes):
        loop.load(st["latents"], st["mask"], st["masked_latents"], st["ctx"])
        sp = loop.prog.stream.cuda_stream
        ev0.record(sp)
        loop.run()
        ev1.record(sp)
        torch.cuda.synchronize()
        gpu_ms += ev0.elapsed_ms(ev1)
    dt = time.perf_counter() - t0
    step_ms = gpu_ms / (passes * T)
    tf = ALGO_TFLOP_PER_CALL.get(latent)
    out = {"workload": f"{'FlintstonesSV' if L == 91 else 'PororoSV'} stage-2, {latent * 8}x{latent * 8}, {T}-step DDIM, CFG {guidance}, "
                       f"batch={S} story x 5 frames, ctx {L}x768" + (", context rows as the reference's builders produce them "
                       "(2 dense images + 8 with L identical rows: SURVEY F6)" if structure == "reference" else ""),
           "passes": passes, "ms_per_step": round(1e3 * dt / passes, 3), "value": round(5 * S * passes / dt, 4),
           "unit": "story-frames/s", "avg_launch_ms": round(step_ms, 4),
           "frac": round(tf * S / (step_ms * 1e-3) / MFMA_F16_PEAK_TFLOPS, 4) if tf else None,
           "shared_cfg_prefix": bool(loop.shared), "rank1_context_plan": loop.rank1_runs is not None,
           "kernels_per_step": loop.prog.plan.n_launch}
    del loop
    return out


def time_prior(passes, sample_steps=50, guidance=4.0):
    """BASELINE config 5 (stage-1 frame-prior transformer, 50-step UnCLIP, 5 frames with CFG) — tools/bench_prior.py's
    measurement, 1 warm-up + `passes` timed stories."""
    import re
    from rcdms_amd import hip
    from rcdms_amd.sampler import PriorLoop
    from rcdms_amd.scheduler import UnCLIPScheduler
    from src.models.myprior_transformer import MyPriorTransformer
    dev = torch.device("cuda", torch.cuda.current_device())
    mk = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
              temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)
    with torch.device("meta"):
        m = MyPriorTransformer(num_attention_heads=32, attention_head_dim=64, num_layers=20, embedding_dim=1280,
                               num_embeddings=91, additional_embeddings=6, unet_use_cross_frame_attention=False,
                               unet_use_temporal_attention=False, use_motion_module=True, motion_module_type="Vanilla",
                               motion_module_kwargs=mk)
    m = m.to_empty(device=dev).eval()
    init_weights_(m)
    nparam = sum(p.numel() for p in m.parameters())
    B, T, E = 10, 91, 1280
    g = torch.Generator(device=dev).manual_seed(42)
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)
    mask = torch.ones(B, T, device=dev)
    mask[:, 20:] = 0
    loop = PriorLoop(m, 5, T, guidance, UnCLIPScheduler(), sample_steps)
    args = (rn(B, E), rn(B, T, E), rn(B, E), rn(B, E), mask)
    lat = rn(5, E)
    loop.load(lat, *args, generator=g)
    loop.run()
    torch.cuda.synchronize()
    ev0, ev1 = hip.Event(), hip.Event()
    gpu_ms, t0 = 0.0, time.perf_counter()
    for _ in range(passes):
        loop.load(lat, *args, generator=g)
        sp = loop.stream.cuda_stream
        ev0.record(sp)
        loop.run()
        ev1.record(sp)
        torch.cuda.synchronize()
        gpu_ms += ev0.elapsed_ms(ev1)
    dt = time.perf_counter() - t0
    step_ms = gpu_ms / (passes * sample_steps)
    flops = 0.0
    for tag in loop.prog.plan.tags:      # algorithmic work of one step, from the launch plan's own descriptors
        g_ = re.match(r"gemm M=(\d+) N=(\d+) K=(\d+)", tag)
        f_ = re.match(r"flash_attn_masked B=(\d+) H=(\d+) L=(\d+) d=(\d+)", tag)
        if g_:
            flops += 2.0 * int(g_.group(1)) * int(g_.group(2)) * int(g_.group(3))
        elif f_:
            bb, hh, ll, dd = (int(v) for v in f_.groups())
            flops += 4.0 * bb * hh * ll * ll * dd
    out = {"workload": f"stage-1 prior transformer, 20 layers x (block + motion module), {nparam / 1e9:.2f} G parameters, batch 10 x 97 "
                       f"tokens, {sample_steps}-step UnCLIP, CFG {guidance}, f16",
           "passes": passes, "ms_per_step": round(1e3 * dt / passes, 3), "value": round(passes / dt, 4), "unit": "stories/s",
           "avg_launch_ms": round(step_ms, 4), "tflop_per_step": round(flops / 1e12, 3),
           "frac": round(flops / (step_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}
    del loop, m
    return out


def extra_configs(model, passes=3):
    """The other BASELINE.json configurations the headline does not carry, timed on the SAME box right after the headline loop
    (so that a driver's run witnesses them): `value` / `ms_per_step` over `passes` full loops each, `frac` = algorithmic
    TFLOP per UNet call / HIP-event step time / 2500.  Never the headline: SURVEY section 8(d) fixes that as config 2 with a
    dense N(0,1) context."""
    out = {}
    # BASELINE configs[0] shape on the GPU: 256x256, 20-step DDIM, one story (the reference's CPU-runnable case)
    out["config1_256x256_20step"] = time_stage2(model, 1, 32, 85, 20, passes)
    # BASELINE configs[2]: FlintstonesSV, 4 stories per batch (b = 8), L = 91
    out["config3_flintstones_batch4"] = time_stage2(model, 4, 64, 91, 50, passes)
    # configs[1] again with the context-row structure the reference's own builders produce (rank-1-context plan)
    out["config2_reference_context_rows"] = time_stage2(model, 1, 64, 85, 50, passes, structure="reference")
    model._invalidate()           # (drop the secondary plans' packed weights and buffers before the prior is built)
    torch.cuda.empty_cache()
    # BASELINE configs[4]: stage-1 frame-prior transformer, 50-step sampling
    out["config5_stage1_prior"] = time_prior(passes)
    torch.cuda.empty_cache()
    return out


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on the process's stdout (fd 1) when its first communicator comes up; the contract is
    ONE JSON line there, so fd 1 points at stderr while communicators are created and first used."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed passes (each = one full T-step story batch)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--stories", type=int, default=1, help="stories per GPU per pass (batch)")
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--ctx-len", type=int, default=85)
    ap.add_argument("--guidance", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the secondary BASELINE configurations (256x256 / 20 steps, FlintstonesSV batch 4, the stage-1 prior, "
                         "the reference's context-row structure) that a default 1-GPU run times after the headline loop")
    ap.add_argument("--context", choices=("dense", "reference"), default="dense",
                    help="context rows of the synthetic story: dense N(0,1) (the headline, SURVEY 8d) or the structure the "
                         "reference's context builders produce (2 dense images + 8 with identical rows, SURVEY F6; not the headline)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-share-prefix", action="store_true",
                    help="evaluate the part of the UNet ahead of the first cross-attention for BOTH CFG halves (A/B switch; "
                         "the default evaluates it once: the halves' inputs are identical there and the result is exact)")
    ap.add_argument("--cfg-split", action="store_true",
                    help="latency mode (not the headline metric): two GPUs per story, one CFG half each, noise predictions "
                         "all-gathered over RCCL inside the step graph; with --gpus 1 ONE half is timed against a one-rank "
                         "communicator (what a rank of a pair does per step, the exchange being a local copy)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: every rank runs on device 0 and the process group is gloo — the N > 1 code path of this "
                         "file (per-rank build, weight broadcast, sharded stories, barrier / max-over-ranks timing) on a box "
                         "with ONE GPU; the printed line is marked and is not a measurement")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="N > 1: seconds any phase with a collective in it (build barrier, weight broadcast, warm-up, timed region, "
                         "gather) may take before this rank reports and exits 86 (rcdms_amd.dist.Watchdog); 0 disables")
    ap.add_argument("--ragged", action="store_true",
                    help="TEST ONLY (with --gpus N > 1): odd ranks hand NO finished story to the final gather, so its uneven-shard "
                         "padding path runs on real device tensors")
    ap.add_argument("--width", type=int, default=0,
                    help="TEST ONLY: a UNet of this base width (block_out_channels = w, 2w, 4w, 4w; cross-attention dim w) instead "
                         "of the 1276.9 M-parameter one, so that N ranks' plans fit ONE device under --share-gpu; the line is INVALID")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="TEST ONLY: exercise the launch / barrier / max-over-ranks harness on CPU (gloo) with a sleep "
                         "in place of the denoising loop; the printed line is marked data=stub and is not a measurement")
    return ap.parse_args(argv)


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks ourselves, one process per device, the way the
    reference's driver does (stage2_batchtest_rcdms_model.py:457-468 spawns one process per GPU) — through
    torch.distributed.run on 127.0.0.1 so every rank sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    import socket
    import subprocess
    if not a.stub_cpu and not a.share_gpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit(f"bench.py: {a.gpus} ranks requested (--gpus {a.gpus}) but {have} device(s) visible; "
                             f"one MI355X per rank is required — refusing to report n_gpus={a.gpus} from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    raise SystemExit(subprocess.call(cmd, env=env))


def timed_passes(one_pass, steps, dist_mod, sync, device):
    """The contract's timed region: barrier + sync, EXACTLY `steps` passes, sync + barrier; returns (max over ranks of
    the wall time, list of every rank's own wall time in rank order)."""
    if dist_mod is not None:
        dist_mod.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    sync()
    if dist_mod is not None:
        dist_mod.barrier()
    sync()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if dist_mod is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        every = [torch.zeros_like(tt) for _ in range(dist_mod.get_world_size())]
        dist_mod.all_gather(every, tt)
        per_rank = [float(x.item()) for x in every]
        dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, per_rank


def traffic_record(S, latent, guidance):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/collect_profiles.sh) — bench.py cannot run
    the profiler on itself.  Returned with its provenance; dropped (None) when the record was taken from a different
    kernel library than the one loaded now, so a stale constant is never passed off as this run's traffic."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not (os.path.exists(tpath) and S == 1 and latent == 64 and guidance > 1):
        return None, None
    try:
        with open(tpath) as f:
            rec = json.load(f)
        from rcdms_amd import build as rbuild
        src = f"profiles/hbm_traffic.json@{rec.get('commit', 'unknown')}"
        if rec.get("csrc_stamp") != rbuild._stamp():   # an unstamped record counts as stale too
            return None, src + " (stale: kernels changed since; not reported)"
        return float(rec["hbm_bytes_per_unet_step"]), src
    except Exception:
        return None, None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a, argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist_on = world > 1
    dist = None
    if a.stub_cpu:
        dev = torch.device("cpu")
        if dist_on:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        dt, per_rank = timed_passes(lambda: time.sleep(0.02 * (rank + 1)), a.steps, dist, lambda: None, dev)
        if rank == 0:
            print(json.dumps({"metric": "harness self-test (no measurement)", "value": 5 * a.stories * a.steps * world / dt,
                              "unit": "story-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "none", "data": "stub",
                              "per_rank_ms": [round(1e3 * x / a.steps, 3) for x in per_rank],
                              "devices": [f"rank{r}:cpu" for r in range(world)]}), flush=True)
        if dist_on:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if a.share_gpu:
        local_rank = 0      # (test harness: all ranks on one device)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if a.share_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()      # brings the RCCL communicator up here, banner and all

    import __graft_entry__
    from rcdms_amd import build as rbuild
    from rcdms_amd.dist import Watchdog
    # the in-tree library is built (or found up to date: content stamp) ONCE per node by the node's first rank; every other
    # rank waits, then checks that what it is about to load belongs to these sources — also the whole story on a node without
    # hipcc, where the shipped rcdms_amd/lib/ must carry the right stamp (rcdms_amd.build.verify)
    node_first = int(os.environ.get("LOCAL_RANK", "0")) == 0
    if node_first:
        __graft_entry__.build()
    if dist_on:
        with Watchdog(a.watchdog, f"rank {rank}: build barrier"):
            dist.barrier()
    if not node_first:
        rbuild.verify()
    from rcdms_amd import hip, synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler

    model = build_model(dev, a.width)
    if dist_on:
        from rcdms_amd.dist import broadcast_module
        # RCCL over xGMI: every replica holds rank 0's weights; matrices travel as f16 (what the kernels consume): 2.55 GB
        with Watchdog(a.watchdog, f"rank {rank}: weight broadcast"):
            broadcast_module(model, src=0, wire_dtype=torch.float16)
            torch.cuda.synchronize()
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    S, T = a.stories, a.ddim_steps
    cdim = a.width or 768
    story = synth.synthetic_story(stories=S, latent_hw=(a.latent, a.latent), ctx_len=a.ctx_len, ctx_dim=cdim, seed=42 + rank,
                                  structure=a.context, cfg=a.guidance > 1)      # (--guidance <= 1: no unconditional half)
    if dist_on and a.guidance > 1:
        # the context every rank shares — the unconditional (empty-prompt) rows of the CFG batch are the same text for every
        # story — is built once on rank 0 and broadcast over RCCL / xGMI (north_star: "RCCL broadcast of the shared
        # reference/text context"); each rank keeps its own conditional rows.  Outside the timed region.
        from rcdms_amd.dist import broadcast_context
        n_u = S * 5
        shared = story["ctx"][:n_u].to(torch.device("cpu") if a.share_gpu else dev).contiguous()
        broadcast_context(shared, src=0)
        story["ctx"] = torch.cat([shared.cpu(), story["ctx"][n_u:]])
    split, units = None, world          # units = independent story batches in flight across the job
    if a.cfg_split:
        from rcdms_amd.dist import CfgSplit
        if world == 1:
            with stdout_to_stderr():
                comm = hip.Comm(hip.Comm.unique_id(), 1, 0)
            split = CfgSplit(0, lambda send, recv, n: comm.allgather(send, recv, n))
        else:
            with stdout_to_stderr():
                split, units = CfgSplit.from_world(), world // 2
            story = synth.synthetic_story(stories=S, latent_hw=(a.latent, a.latent), ctx_len=a.ctx_len, ctx_dim=cdim,
                                          seed=42 + rank // 2, structure=a.context)
    loop = DenoiseLoop(model, S, 5, a.latent, a.latent, a.ctx_len, a.guidance, sched, T,
                       share_cfg_prefix=not a.no_share_prefix, cfg_split=split)

    def one_pass():
        loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
        loop.run(use_graph=not a.no_graph)

    with Watchdog(a.watchdog if dist_on else 0, f"rank {rank}: warm-up passes"):
        for _ in range(a.warmup):
            one_pass()
        torch.cuda.synchronize()

    # timed region: inputs are re-staged (a few MB H2D) inside load(); the loop itself is K x T graph replays
    ev0, ev1 = hip.Event(), hip.Event()
    gpu_ms = [0.0]

    def timed_pass():
        loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
        sp = loop.prog.stream.cuda_stream
        ev0.record(sp)
        loop.run(use_graph=not a.no_graph)
        ev1.record(sp)
        gpu_ms[0] += ev0.elapsed_ms(ev1)

    with Watchdog(a.watchdog if dist_on else 0, f"rank {rank}: timed region ({a.steps} passes)"):
        dt, per_rank = timed_passes(timed_pass, a.steps, dist if dist_on else None, torch.cuda.synchronize,
                                    torch.device("cpu") if a.share_gpu else dev)   # (gloo gathers host tensors)

    loop_rank1 = loop.rank1_runs is not None
    frames = 5 * S * a.steps * units
    value = frames / dt
    launches = a.steps * T
    avg_launch_ms = gpu_ms[0] / launches
    tf_call = ALGO_TFLOP_PER_CALL.get(a.latent) if not a.width else None
    traffic, traffic_src = traffic_record(S, a.latent, a.guidance) if not a.cfg_split else (None, None)
    roof = None
    if tf_call is not None:
        achieved = tf_call * S / (avg_launch_ms * 1e-3) * (0.5 if a.cfg_split else 1.0)   # a split rank evaluates one CFG half
        roof = {"bound": "mfma", "kernel": "denoise-step graph (UNet b=%d + CFG + DDIM)" % (2 * S if a.guidance > 1 and not a.cfg_split else S),
                "achieved": round(achieved, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_launch_ms, 4), "launches": launches}
        # `achieved` prices the REFERENCE's flops per call; what this build does not issue (the CFG halves' identical prefix,
        # evaluated once) is reported beside it so that the utilisation of the work actually launched can be read off too
        skipped = SHARED_PREFIX_TFLOP.get(a.latent, 0.0) if loop.shared else 0.0
        # ... and the 5/9 of every Upsample3D conv that the phase form (rcdm_conv3x3 upsample = 2) does not multiply: exact
        # algebra (a nearest-2x upsampled pixel grid holds every source pixel four times), read off the launch plan
        up2 = upsample_phase_tflop(loop.prog.plan) / S
        up9 = upsample_taps_tflop(loop.prog.plan) / S      # (round 6: nine tap planes over the source pixels + gather, 27 of 36 left out)
        # ... and the 20/36 of the ResNet 3x3 convolutions' multiply-adds the Winograd F(2x2, 3x3) form does not issue (round 6)
        wino = winograd_tflop(loop.prog.plan) / S
        roof["flops_skipped_tflop_per_launch"] = round((skipped + up2 + up9 + wino) * S, 4)
        roof["flops_skipped_breakdown"] = {"shared_cfg_prefix": round(skipped * S, 4), "upsample_phase_form": round(up2 * S, 4),
                                           "upsample_tap_planes": round(up9 * S, 4), "winograd_form": round(wino * S, 4)}
        roof["achieved_issued"] = round(achieved * (1.0 - (skipped + up2 + up9 + wino) / (tf_call * (0.5 if a.cfg_split else 1.0))), 1)

    out = {
        "metric": "story-frames/sec (stage-2 UNet, 50-step DDIM, 512^2)", "value": round(value, 4),
        "unit": "story-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{'FlintstonesSV' if a.ctx_len == 91 else 'PororoSV'} stage-2, {a.latent * 8}x{a.latent * 8}, "
                               f"{T}-step DDIM, CFG {a.guidance}, "
                               f"batch={S} story x 5 frames per GPU, ctx {a.ctx_len}x768, random-init 1276.9M-param UNet3D",
                   "stories_per_gpu": S, "latent": a.latent, "ddim_steps": T, "parallelism": (f"story-replicas x{world}" if not a.cfg_split else
                                   "cfg-split: one CFG half timed, one-rank all-gather" if world == 1 else
                                   f"cfg-split pairs x{world // 2} (2 GPUs per story, RCCL all-gather per step)"),
                   "shared_cfg_prefix": bool(loop.shared)},
        "per_rank_ms": [round(1e3 * x / a.steps, 3) for x in per_rank],
        "roofline": roof,
    }
    if dist_on:
        # what the communicator actually spanned: one entry per rank (device index, name, uuid tail), and the RCCL version
        props = torch.cuda.get_device_properties(dev)
        mine = f"rank{rank}:cuda{local_rank}:{props.name}:{str(getattr(props, 'uuid', ''))[-8:]}"
        names = [None] * world
        dist.all_gather_object(names, mine)
        out["devices"] = names
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out["rccl_version"] = None
    if dist_on and not a.cfg_split:
        # finished latents back to rank 0 (the reference's processes write their own PNGs, stage2_batchtest_rcdms_model.py:
        # 378-401; a served job collects them): shard sizes are exchanged and padded, so ragged shards work (--ragged: TEST)
        from rcdms_amd.dist import gather_stories
        mine = loop.lat.detach().clone()
        if a.ragged and rank % 2:
            mine = mine[:0]                      # odd ranks contribute no story: the padding path runs on device tensors
        with Watchdog(a.watchdog, f"rank {rank}: gather of the finished stories"):
            got = gather_stories(mine.cpu() if a.share_gpu else mine, dst=0)
        if rank == 0:
            want = sum(0 if (a.ragged and r % 2) else S for r in range(world))
            assert got is not None and got.shape[0] == want and bool(torch.isfinite(got).all()), (None if got is None else got.shape, want)
            out["gathered_stories"] = int(got.shape[0])
    if rank == 0 and world == 1 and not a.cfg_split and not a.no_extra_configs and not a.no_graph and a.latent == 64 and S == 1:
        del loop
        out["extra_configs"] = extra_configs(model)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, a.latent, a.ctx_len, T)
    if a.context != "dense":
        out["config"]["context_rows"] = "reference structure (2 dense images + 8 with identical rows, SURVEY F6): NOT the headline workload"
        out["config"]["rank1_context_plan"] = loop_rank1
    if a.share_gpu or a.width or a.ragged:
        out["data"] = "INVALID (--share-gpu / --width / --ragged harness test: not a measurement)"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
