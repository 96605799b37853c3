"""ORACLE — TEST INFRASTRUCTURE ONLY.  Mint the golden vectors under tests/golden/ by running the
REFERENCE's own model classes (imported from /root/reference through oracle/ref_scaffold.py) on CPU fp32
with procedural name-seeded weights (rcdms_amd/synth.py).  Run in the build container only:

    python -m oracle.make_golden [--full] [--only blocks|tiny|ctx|prior|full]   # --full adds the full-width UNet (32x32, 64x64)
    python -m oracle.make_golden --only loop32|loop64|cfg3     # reference-UNet-driven DDIM trajectories / config-3 story
    python -m oracle.make_golden --only eps32|eps64            # reference eps at stored trajectory points x_k (mid / late steps)
    python -m oracle.make_golden --only skewed                 # full-width UNet with the second ("skewed") weight family, 32x32
    python -m oracle.make_golden --only rank1ctx               # context rows 2..9 with L identical rows each, tiny + full width
    python -m oracle.make_golden --only sdlike                 # full-width UNet with the third ("sdlike") weight family, 32x32

What is stored: small inputs and the reference outputs (fp32 .npz), plus a digest of the reference's
state-dict key/shape list so the mirrored classes are checked to have the identical 1286-key layout.
Weights are NOT stored: both sides regenerate them from the parameter names."""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_scaffold  # noqa: E402
from rcdms_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def key_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(f"{k}:{tuple(sd[k].shape)};".encode())
    return h.hexdigest()


def load_procedural(module, seed, style="unit"):
    sd = module.state_dict()
    new = synth.procedural_state_dict({k: v.shape for k, v in sd.items()}, seed, style)
    module.load_state_dict(new)
    return key_digest(sd)


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


@torch.no_grad()
def blocks():
    ref_scaffold.load_reference_models()
    import refsrc.models.attention as r_att
    import refsrc.models.motion_module as r_mm
    import refsrc.models.resnet as r_res

    # ResnetBlock3D, Cin != Cout (1x1 shortcut) and Cin == Cout — cross-frame GroupNorm (F2)
    for tag, cin, cout in (("resnet_64_128", 64, 128), ("resnet_64_64", 64, 64)):
        m = r_res.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=256, eps=1e-5, groups=32,
                                non_linearity="silu", use_inflated_groupnorm=False).eval()
        dig = load_procedural(m, seed=1)
        x, temb = randn((2, cin, 5, 8, 8), 10), randn((2, 256), 11)
        save(tag, x=x, temb=temb, y=m(x, temb), digest=dig)

    # Transformer3DModel: 8 heads x 8, context 13 tokens x 64
    m = r_att.Transformer3DModel(8, 8, in_channels=64, num_layers=1, cross_attention_dim=64, norm_num_groups=32,
                                 use_linear_projection=False, upcast_attention=False,
                                 unet_use_cross_frame_attention=False, unet_use_temporal_attention=False).eval()
    dig = load_procedural(m, seed=2)
    x, ctx = randn((2, 64, 5, 8, 8), 20), randn((10, 13, 64), 21)
    save("transformer_64", x=x, ctx=ctx, y=m(x, encoder_hidden_states=ctx).sample, digest=dig)

    # VanillaTemporalModule (weights procedural, so the zero-initialised proj_out is NOT zero here)
    m = r_mm.VanillaTemporalModule(in_channels=64, **ref_scaffold.TESTING_YAML_UNET_KWARGS["motion_module_kwargs"]).eval()
    dig = load_procedural(m, seed=3)
    x = randn((2, 64, 5, 8, 8), 30)
    save("motion_64", x=x, y=m(x, None, None), digest=dig)

    # samplers
    m = r_res.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op").eval()
    dig = load_procedural(m, seed=4)
    x = randn((2, 64, 5, 8, 8), 40)
    save("downsample_64", x=x, y=m(x), digest=dig)
    m = r_res.Upsample3D(64, use_conv=True, out_channels=64).eval()
    dig = load_procedural(m, seed=5)
    save("upsample_64", x=x, y=m(x), digest=dig)
    m = r_res.InflatedConv3d(9, 64, kernel_size=3, padding=(1, 1)).eval()
    dig = load_procedural(m, seed=6)
    x9 = randn((2, 9, 5, 8, 8), 41)
    save("conv_in_9_64", x=x9, y=m(x9), digest=dig)


@torch.no_grad()
def tiny_unet():
    """Full stage-2 topology at width 64 (layers_per_block 2): 16x16 and 32x32 latents."""
    m = ref_scaffold.build_reference_unet(width=64, cross_dim=64)
    dig = load_procedural(m, seed=7)
    print("  tiny UNet params: %.1f M, keys %d" % (sum(p.numel() for p in m.parameters()) / 1e6, len(m.state_dict())))
    for hw, t in ((16, 981), (32, 441)):
        x, ctx = randn((2, 9, 5, hw, hw), 50 + hw), randn((10, 13, 64), 51 + hw)
        y = m(x, torch.tensor(t), encoder_hidden_states=ctx, return_dict=False)[0]
        save(f"unet_tiny_{hw}", x=x, ctx=ctx, t=np.int64(t), y=y, digest=dig)
    # per-batch timesteps tensor + return_dict=True path returns the raw tensor (unet.py:462)
    x, ctx = randn((2, 9, 5, 16, 16), 60), randn((10, 13, 64), 61)
    y = m(x, torch.tensor([981, 1]), ctx)
    assert torch.is_tensor(y)
    save("unet_tiny_16_tvec", x=x, ctx=ctx, t=np.array([981, 1], dtype=np.int64), y=y, digest=dig)


@torch.no_grad()
def full_unet():
    """Full-width (1276.9 M parameter) reference UNet on the synthetic story of SURVEY §8(d): outputs only."""
    t0 = time.time()
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=0)
    print("  full UNet built + procedural weights in %.0f s; keys %d" % (time.time() - t0, len(m.state_dict())))
    for hw, t in ((32, 951), (64, 981)):
        s = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=42)
        x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)
        t0 = time.time()
        y = m(x, torch.tensor(t), encoder_hidden_states=s["ctx"], return_dict=False)[0]
        print("  reference forward %dx%d: %.1f s" % (hw, hw, time.time() - t0))
        save(f"unet_full_{hw}", t=np.int64(t), y=y, digest=dig)


ALT_SEED, ALT_STORY_SEED, ALT_T = 5, 43, 501


@torch.no_grad()
def full_unet_skewed():
    """The full-width reference UNet with the SECOND weight family (synth style "skewed": per-channel log-normal gains,
    heavy-tailed entries, wider norm parameters) on another story at a mid-trajectory timestep, 32x32 and 64x64 latents."""
    t0 = time.time()
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=ALT_SEED, style="skewed")
    print("  full UNet built + skewed weights in %.0f s" % (time.time() - t0))
    for hw in (32, 64):
        s = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=ALT_STORY_SEED)
        x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)
        t0 = time.time()
        y = m(x, torch.tensor(ALT_T), encoder_hidden_states=s["ctx"], return_dict=False)[0]
        print("  reference forward %dx%d (skewed weights): %.1f s; |y| rms %.3f max %.3f"
              % (hw, hw, time.time() - t0, y.pow(2).mean().sqrt(), y.abs().max()))
        save(f"unet_full_{hw}_skewed", t=np.int64(ALT_T), y=y, digest=dig, seed=np.int64(ALT_SEED), story_seed=np.int64(ALT_STORY_SEED))


@torch.no_grad()
def rank1_context():
    """The reference UNet on a context with the row structure its own context builders produce (SURVEY F5 / F6,
    RCDMs_pipeline.py:444-450): rows 0..1 dense (the seen frame of each CFG half), rows 2..9 with L identical rows each (the
    unseen frames: semantic_stack has one key / value token) — synth.synthetic_story(structure="reference").  Full width
    at 32x32 and 64x64 latents (same weights as unet_full_*) and the width-64 topology at 16x16 (with its inputs)."""
    m = ref_scaffold.build_reference_unet(width=64, cross_dim=64)
    dig = load_procedural(m, seed=7)
    s = synth.synthetic_story(stories=1, latent_hw=(16, 16), ctx_len=13, ctx_dim=64, seed=44, structure="reference")
    x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)
    y = m(x, torch.tensor(741), encoder_hidden_states=s["ctx"], return_dict=False)[0]
    save("unet_tiny_16_rank1ctx", x=x, ctx=s["ctx"], t=np.int64(741), y=y, digest=dig)
    t0 = time.time()
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=0)
    print("  full UNet built + procedural weights in %.0f s" % (time.time() - t0))
    for hw, t in ((32, 951), (64, 981)):
        s = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=42, structure="reference")
        x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)
        t0 = time.time()
        y = m(x, torch.tensor(t), encoder_hidden_states=s["ctx"], return_dict=False)[0]
        print("  reference forward %dx%d (rank-1 context rows 2..9): %.1f s" % (hw, hw, time.time() - t0))
        save(f"unet_full_{hw}_rank1ctx", t=np.int64(t), y=y, digest=dig)


SD_SEED, SD_STORY_SEED, SD_T = 9, 45, 681


@torch.no_grad()
def full_unet_sdlike():
    """The full-width reference UNet with the THIRD weight family (synth style "sdlike": 50-100x outlier channels feeding the
    GroupNorms, 25x attention logits) at 32x32 latents; also stores the largest activation the reference sees on its
    residual stream (max |conv_norm_out input|) so the report of the f16 path can be read against it."""
    t0 = time.time()
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=SD_SEED, style="sdlike")
    print("  full UNet built + sdlike weights in %.0f s" % (time.time() - t0))
    seen = {}
    m.conv_norm_out.register_forward_hook(lambda mod, inp, out: seen.__setitem__("final", float(inp[0].abs().max())))
    s = synth.synthetic_story(stories=1, latent_hw=(32, 32), ctx_len=85, seed=SD_STORY_SEED)
    x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)
    t0 = time.time()
    y = m(x, torch.tensor(SD_T), encoder_hidden_states=s["ctx"], return_dict=False)[0]
    print("  reference forward 32x32 (sdlike weights): %.1f s; |y| rms %.3f max %.3f; max |residual stream| %.1f"
          % (time.time() - t0, y.pow(2).mean().sqrt(), y.abs().max(), seen["final"]))
    save("unet_full_32_sdlike", t=np.int64(SD_T), y=y, digest=dig, seed=np.int64(SD_SEED), story_seed=np.int64(SD_STORY_SEED),
         max_final=np.float32(seen["final"]))


@torch.no_grad()
def ctx_stacks():
    """The driver's fine_stack / semantic_stack (stage2_batchtest_rcdms_model.py:117-149) at the shapes the pipeline
    feeds them for mask [1,0,0,0,0] with CFG: 2 seen rows x 257 patch tokens x 1664, 8 unseen rows x 1 x 1280."""
    fine_cls, sem_cls = ref_scaffold.load_reference_context_stacks()
    for name, cls, vis_dim, k, lv, seed in (("ctx_fine", fine_cls, 1664, 2, 257, 301),
                                            ("ctx_semantic", sem_cls, 1280, 8, 1, 302),
                                            ("ctx_fine_ragged", fine_cls, 1664, 3, 70, 303)):
        m = cls(text_dim=768, vis_dim=vis_dim).eval()
        digest = load_procedural(m, seed)
        vis = synth.normal_tensor(name + ".vis", (k, lv, vis_dim), seed)   # inputs are regenerated by the tests
        text = synth.normal_tensor(name + ".text", (k, 85, 768), seed)
        out = m(vis, text)
        save(name, out=out, seed=seed, k=k, lv=lv, vis_dim=vis_dim, key_digest=digest)


PRIOR_CASES = {  # name: (num_layers, heads, head_dim, embedding_dim, seed)
    "prior_tiny": (2, 4, 64, 128, 401),
    "prior_full": (20, 32, 64, 1280, 402),   # the Kandinsky-2.2 prior shape the driver loads (2.85 G parameters)
}


def prior_inputs(name, B, E, T, seed):
    """Inputs of MyPriorTransformer.forward for one CFG batch of a 5-frame story (B = 2 x 5), regenerated by the tests."""
    t = lambda k, shape: synth.normal_tensor(f"{name}.{k}", shape, seed)
    am = torch.ones(B, T)
    for b in range(B):
        am[b, 12 + 3 * b:] = 0.0          # text padding: tokens past the caption length are masked
    return dict(hidden_states=t("hidden_states", (B, E)), proj_embedding=t("proj_embedding", (B, E)),
                encoder_hidden_states=t("encoder_hidden_states", (B, T, E)), proj_embedding1=t("proj_embedding1", (B, E)),
                mask_label=t("mask_label", (B, E)), attention_mask=am)


@torch.no_grad()
def prior(which):
    """Stage-1 prior transformer: outputs of the reference class on procedural weights."""
    for name in which:
        layers, heads, hd, E, seed = PRIOR_CASES[name]
        t0 = time.time()
        m = ref_scaffold.build_reference_prior(num_layers=layers, heads=heads, head_dim=hd, embedding_dim=E)
        digest = load_procedural(m, seed)
        x = prior_inputs(name, 10, E, 91, seed)
        y = m(x["hidden_states"], torch.tensor(481), x["proj_embedding"], x["encoder_hidden_states"],
              x["proj_embedding1"], x["mask_label"], attention_mask=x["attention_mask"]).predicted_image_embedding
        print("  reference prior %s: %.1f s" % (name, time.time() - t0))
        save(name, y=y, t=np.int64(481), seed=seed, key_digest=digest,
             cfg=np.array([layers, heads, hd, E], dtype=np.int64))


LOOP_KEEP = {32: list(range(1, 21)), 64: [1, 2, 3, 5, 10, 11, 20, 21, 30, 31, 40, 41, 49, 50]}


@torch.no_grad()
def loop_trajectory(hw, steps):
    """BASELINE config 1 (32x32 latents, 20 steps) / config 2 (64x64, 50 steps): the REFERENCE UNet3DConditionModel
    (full width, procedural weights, seed-42 synthetic story) driven through the oracle's restatement of the pipeline
    loop (RCDMs_pipeline.py:455-503: CFG 2.0, DDIM).  Stores the latents after selected steps, so the HIP loop can be
    checked per step (x_k -> x_k+1 from a stored x_k) and end to end.  ~12 s (32x32) / ~70 s (64x64) per step here."""
    from oracle import unet_oracle as O
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=0)
    s = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=42)
    keep, out, t0 = set(LOOP_KEEP[hw]), {}, time.time()

    def unet(x, t, ctx):
        return m(x, torch.tensor(int(t)), encoder_hidden_states=ctx, return_dict=False)[0]

    def cb(i, t, x):
        if i + 1 in keep:
            out[f"x{i + 1}"] = x.clone()
        print("    step %d/%d t=%d  |x| rms %.4f  (%.0f s)" % (i + 1, steps, t, x.pow(2).mean().sqrt(), time.time() - t0),
              flush=True)
        if (i + 1) % 10 == 0 or i + 1 == steps:      # checkpoint: a partial file is still a usable fixture
            save(f"loop_full_{hw}", steps=np.int64(steps), done=np.int64(i + 1), guidance=np.float32(2.0), digest=dig, **out)

    O.denoise_loop(None, None, s["latents"], s["mask"], s["masked_latents"], s["ctx"], steps, 2.0, unet=unet, callback=cb)


EPS_AT = {32: (5, 10, 19), 64: (10, 30, 49)}


@torch.no_grad()
def eps_along_trajectory(hw):
    """The reference UNet's raw output eps at points x_k of the stored reference trajectory (loop_full_<hw>.npz), k in the
    middle and at the end of the loop, where |x| has grown ~20x over x_0 (the f16 headroom case VERDICT r3 asks for).  The
    UNet input at loop index k is cat([x_k]*2, mask, masked) at t_k = timesteps[k] (RCDMs_pipeline.py:455-476)."""
    from oracle import unet_oracle as O
    g = np.load(os.path.join(GOLD, f"loop_full_{hw}.npz"))
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=0)
    assert dig == str(g["digest"])
    s = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=42)
    sched = O.DDIMOracle()
    sched.set_timesteps(int(g["steps"]))
    out = {}
    for k in EPS_AT[hw]:
        xk = torch.from_numpy(g[f"x{k}"])
        t = int(sched.timesteps[k])
        x = torch.cat([torch.cat([xk] * 2), s["mask"], s["masked_latents"]], dim=1)
        t0 = time.time()
        y = m(x, torch.tensor(t), encoder_hidden_states=s["ctx"], return_dict=False)[0]
        print("  reference eps at k=%d t=%d: |x| rms %.3f, |eps| rms %.3f max %.3f  (%.0f s)"
              % (k, t, xk.pow(2).mean().sqrt(), y.pow(2).mean().sqrt(), y.abs().max(), time.time() - t0), flush=True)
        out[f"eps{k}"] = y
        out[f"t{k}"] = np.int64(t)
    save(f"eps_full_{hw}", ks=np.array(EPS_AT[hw], dtype=np.int64), digest=dig, **out)


@torch.no_grad()
def config3_story():
    """BASELINE config 3 (FlintstonesSV, L = 91, 4 stories = b 8, 64x64 latents): the reference UNet run on ONE story
    of the seed-44 four-story batch (its two CFG rows, b = 2) — the batch itself is 4x that and stories are independent."""
    m = ref_scaffold.build_reference_unet()
    dig = load_procedural(m, seed=0)
    s = synth.synthetic_story(stories=4, latent_hw=(64, 64), ctx_len=91, seed=44)
    i = 2
    rows = [i, 4 + i]
    x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1)[rows]
    ctx = s["ctx"].view(8, 5, 91, 768)[rows].reshape(10, 91, 768)
    t0 = time.time()
    y = m(x, torch.tensor(961), encoder_hidden_states=ctx, return_dict=False)[0]
    print("  reference forward config-3 story %d: %.1f s" % (i, time.time() - t0))
    save("unet_full_64_cfg3", t=np.int64(961), story=np.int64(i), y=y, digest=dig)


def pretrained_2d():
    """UNet3DConditionModel.from_pretrained_2d (reference unet.py:465-509) run on a tiny SD-style 2-D checkpoint folder
    (rcdms_amd.synth.write_2d_checkpoint).  Stored: which keys the reference leaves missing / reports unexpected, the
    config fields it ends up with, and a float64 checksum per loaded tensor group — the mirrored classmethod must agree."""
    import io
    import tempfile
    from contextlib import redirect_stdout
    ref_unet = ref_scaffold.load_reference_models()
    kw = ref_scaffold.TESTING_YAML_UNET_KWARGS
    cfg9 = dict(synth.TINY_2D_CONFIG, in_channels=9)
    shapes = {k: tuple(v.shape) for k, v in ref_unet.UNet3DConditionModel.from_config(
        {k: v for k, v in cfg9.items() if not k.endswith("block_types")}, **kw).state_dict().items()}
    with tempfile.TemporaryDirectory() as d:
        file_sd = synth.write_2d_checkpoint(os.path.join(d, "unet"), shapes)
        buf = io.StringIO()
        with redirect_stdout(buf):
            m = ref_unet.UNet3DConditionModel.from_pretrained_2d(d, subfolder="unet", unet_additional_kwargs=kw)
    printed = buf.getvalue()
    sd = m.state_dict()
    loaded = sorted(k for k in file_sd if k in sd and not k.startswith("conv_in"))
    missing = sorted(k for k in sd if k not in file_sd or k.startswith("conv_in"))
    unexpected = sorted(k for k in file_sd if k not in sd and not k.startswith("conv_in"))
    assert f"### missing keys: {len(missing)}" in printed and f"### unexpected keys: {len(unexpected)}" in printed, printed
    for k in loaded:
        assert torch.equal(sd[k], file_sd[k]), k
    csum = sum(sd[k].double().sum().item() for k in loaded)
    n_temporal = sum(p.numel() for n, p in m.named_parameters() if "temporal" in n)
    print(f"  reference: {len(loaded)} loaded, {len(missing)} missing, {len(unexpected)} unexpected; conv_in {tuple(sd['conv_in.weight'].shape)}")
    save("from_pretrained_2d", missing=np.array("\n".join(missing)), unexpected=np.array("\n".join(unexpected)), n_loaded=np.int64(len(loaded)),
         checksum=np.float64(csum), n_temporal=np.int64(n_temporal), conv_in_shape=np.array(sd["conv_in.weight"].shape),
         in_channels=np.int64(m.config.in_channels), down0=np.array(str(m.config.down_block_types[0])),
         digest=key_digest(sd))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count())))
    if a.only in ("", "blocks"):
        print("blocks"); blocks()
    if a.only in ("", "tiny"):
        print("tiny UNet"); tiny_unet()
    if a.only in ("", "ctx"):
        print("context stacks"); ctx_stacks()
    if a.only in ("", "prior"):
        print("prior transformer"); prior(["prior_tiny"] + (["prior_full"] if a.full else []))
    if a.full or a.only == "full":
        print("full UNet"); full_unet()
    if a.only == "rank1ctx":
        print("rank-1 context rows"); rank1_context()
    if a.only == "sdlike":
        print("full UNet, sdlike weights"); full_unet_sdlike()
    if a.only == "skewed":
        print("full UNet, second weight family"); full_unet_skewed()
    if a.only == "pretrained2d":
        print("from_pretrained_2d"); pretrained_2d()
    if a.only == "loop32":
        print("config-1 trajectory"); loop_trajectory(32, 20)
    if a.only == "loop64":
        print("config-2 trajectory"); loop_trajectory(64, 50)
    if a.only == "cfg3":
        print("config-3 story"); config3_story()
    if a.only in ("eps32", "eps64"):
        print("eps along the trajectory"); eps_along_trajectory(int(a.only[3:]))
