"""The stage-2 denoising loop on MI355X: T replays of ONE captured hipGraph per story batch.

Reference: the `for t in timesteps` body of RCDMsPipeline.__call__ (src/pipelines/RCDMs_pipeline.py:480-503):
cat([latents]*2) -> cat([x, mask, masked_latents], 1) -> unet -> CFG combine -> scheduler.step.  Here one graph
holds [this step's time_emb_proj rows from a per-schedule table | assemble the 9-channel rows | ~10^3 UNet kernels |
fused CFG+DDIM update | step++]; the step index
and the coefficient table live in device memory, so the host issues exactly one hipGraphLaunch per step and never
touches a parameter.  Generalised over the reference's hard-coded batch 1 / 64x64 (:408,:476) to S stories."""
import torch

from . import hip
from . import switches as SW
from .engine import CIN_PAD, full_rank_runs


class DenoiseLoop:
    def __init__(self, unet, stories, frames, height, width, ctx_len, guidance_scale, scheduler, num_steps,
                 share_cfg_prefix=True, cfg_split=None, rank1_context=True):
        """cfg_split (rcdms_amd.dist.CfgSplit or anything with .half and .allgather(send_ptr, recv_ptr, nbytes)): the
        two-GPUs-per-story latency mode — this rank evaluates only CFG half `half` of the UNet (batch S instead of 2S),
        the halves' noise predictions are all-gathered inside the step graph and every rank applies the same CFG +
        DDIM update.  load() still takes the full CFG batch of mask / masked latents / context."""
        self.unet, self.S, self.f, self.H, self.W = unet, stories, frames, height, width
        self.gs = float(guidance_scale)
        self.reps = 2 if guidance_scale > 1.0 else 1
        self.T = int(num_steps)
        dev = unet.device
        self.device = dev
        if not hasattr(scheduler, "alphas_cumprod"):
            raise NotImplementedError(f"{type(scheduler).__name__}: only DDIM / PNDM schedulers have a fused HIP step")
        cfg = getattr(scheduler, "config", {})
        if getattr(cfg, "prediction_type", "epsilon") != "epsilon":
            raise NotImplementedError("only epsilon prediction")
        # the fused rcdm_cfg_ddim_step is DDIM eq. (12) with eta = 0 and no clipping / thresholding — what the reference
        # pipeline configures (RCDMs_pipeline.py:84-109 forces clip_sample False); anything else must not run silently
        if getattr(cfg, "clip_sample", False) or getattr(cfg, "thresholding", False):
            raise NotImplementedError("DenoiseLoop: clip_sample / thresholding are not built into the fused DDIM step "
                                      "(the reference pipeline sets clip_sample=False)")
        scheduler.set_timesteps(int(num_steps), device=None)
        ts = torch.as_tensor(scheduler.timesteps).to("cpu", torch.int64)
        self.timesteps = ts
        # PNDM (RCDMs_pipeline.py:72-79 accepts it; PLMS form): num_steps + 1 model evaluations, the multistep combination and
        # the update in rcdm_cfg_pndm_step from the scheduler's own per-call table
        self.pndm = hasattr(scheduler, "plms_table")
        self.T = len(ts)
        self._pndm_next = 0   # PNDM only: the schedule row the PLMS history in self.hist is valid for (run() checks it)
        if self.pndm:
            self.coef = scheduler.plms_table().to(dev)
            self.hist = torch.zeros(5, stories * 4 * frames * height * width, dtype=torch.float32, device=dev)
        else:
            # the fused step is DDIM: the timestep list must be DDIM's — strictly decreasing at the constant stride
            # num_train_timesteps // T.  A multistep scheduler without plms_table() (e.g. a diffusers PNDMScheduler object:
            # alphas_cumprod, but N + 1 timesteps with the second one repeated) must not be run with DDIM coefficients.
            ratio = int(cfg.get("num_train_timesteps", 1000)) // self.T
            tl = ts.tolist()
            if getattr(cfg, "skip_prk_steps", None) is not None or any(a - b != ratio for a, b in zip(tl, tl[1:])):
                raise NotImplementedError(
                    f"{type(scheduler).__name__}: its timesteps are not a DDIM schedule of stride {ratio} (a multistep "
                    "scheduler?) — pass rcdms_amd.scheduler.DDIMScheduler or rcdms_amd.scheduler.PNDMScheduler")
            ac = torch.as_tensor(scheduler.alphas_cumprod).double().cpu()
            final = torch.as_tensor(getattr(scheduler, "final_alpha_cumprod", 1.0)).double().cpu()
            rows = []
            for t in ts.tolist():
                prev = t - ratio
                a_t, a_p = ac[t], (ac[prev] if prev >= 0 else final)
                rows.append([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()])
            self.coef = torch.tensor(rows, dtype=torch.float32).to(dev)
        self.ts_dev = ts.to(torch.float32).to(dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.init_noise_sigma = float(getattr(scheduler, "init_noise_sigma", 1.0))

        self.ctx_len = ctx_len
        S, R, f, H, W = stories, self.reps, frames, height, width
        self.lat = torch.zeros(S, 4, f, H, W, dtype=torch.float32, device=dev)
        self.mask = torch.zeros(R * S, 1, f, H, W, dtype=torch.float32, device=dev)
        self.masked = torch.zeros(R * S, 4, f, H, W, dtype=torch.float32, device=dev)
        # Two launch plans, built on first use: the general one, and the "shared prefix" one for the case the reference
        # pipeline always produces under CFG (RCDMs_pipeline.py:481-482: latents, mask and masked latents of the two
        # halves are the same tensors) where the part of the UNet ahead of the first cross-attention is evaluated once.
        self.split = cfg_split
        if cfg_split is not None and self.reps != 2:
            raise ValueError("cfg_split needs classifier-free guidance (guidance_scale > 1)")
        self.share_allowed = share_cfg_prefix and cfg_split is None
        self.rank1_allowed = bool(rank1_context) and SW.RANK1_CTX
        self._variants = {}
        self._v = None

    def _select(self, share, runs=None):
        """share: the shared-CFG-prefix plan; runs: the full-rank image runs of the context (engine.full_rank_runs) when some
        images' context rows are all equal — the rank-1-context plan — else None."""
        v = self._variants.get((share, runs))
        if v is None:
            S, R, f, H, W = self.S, self.reps, self.f, self.H, self.W
            if self.split is not None:
                # one CFG half here: batch rows [half * S, (half + 1) * S) of the reference's cat([latents] * 2)
                half, b = self.split.half, S
                p = self.unet.program(b, f, H, W, self.ctx_len, shared_prefix=False, rank1_runs=runs)
                eps = p.eps_out
                nbytes = eps.M * eps.ld * 2
                self.eps_full = torch.empty(2 * eps.M, eps.ld, dtype=torch.float16, device=self.device)
                m_off = half * S * f * H * W * 4          # fp32 (R*S, 1, f, H, W)
                k_off = half * S * 4 * f * H * W * 4      # fp32 (R*S, 4, f, H, W)
                table = p.time_table(self.timesteps.tolist())
                pre = [
                    lambda: hip.load_table_row(table.data_ptr(), self.step_dev.data_ptr(), p.tproj.data_ptr(), table.shape[1]),
                    lambda: hip.assemble_input(self.lat.data_ptr(), self.mask.data_ptr() + m_off,
                                               self.masked.data_ptr() + k_off, S, 1, f, H, W, p.x_in.ptr, p.x_in.ld, CIN_PAD),
                ]
                post = [
                    lambda: self.split.allgather(eps.ptr, self.eps_full.data_ptr(), nbytes),
                    lambda: self._sched_step(self.eps_full.data_ptr(), eps.ld),
                    lambda: hip.advance_step(self.step_dev.data_ptr()),
                ]
            else:
                b = R * S
                p = self.unet.program(b, f, H, W, self.ctx_len, shared_prefix=share, rank1_runs=runs)
                # the timestep-embedding chain and all time_emb_proj rows for the T steps of the schedule, once
                table = p.time_table(self.timesteps.tolist())
                pre = [
                    lambda: hip.load_table_row(table.data_ptr(), self.step_dev.data_ptr(), p.tproj.data_ptr(), table.shape[1]),
                    lambda: hip.assemble_input(self.lat.data_ptr(), self.mask.data_ptr(), self.masked.data_ptr(), S, R, f, H, W,
                                               p.x_in.ptr, p.x_in.ld, CIN_PAD),
                ]
                post = [
                    lambda: self._sched_step(p.eps_out.ptr, p.eps_out.ld),
                    lambda: hip.advance_step(self.step_dev.data_ptr()),
                ]
            v = self._variants[(share, runs)] = dict(prog=p, pre=pre, post=post, graph=None)
        self.shared = share
        self.rank1_runs = v["prog"].rank1_runs
        self._v = v

    def _sched_step(self, eps_ptr, ld):
        """CFG combine + scheduler.step (RCDMs_pipeline.py:492-497) on the device-resident latents, row `step` of the table."""
        S, R, f, H, W = self.S, self.reps, self.f, self.H, self.W
        if self.pndm:
            hip.cfg_pndm_step(eps_ptr, ld, self.lat.data_ptr(), self.hist.data_ptr(), S, R, f, H, W, self.gs, self.coef.data_ptr(),
                              self.step_dev.data_ptr())
        else:
            hip.cfg_ddim_step(eps_ptr, ld, self.lat.data_ptr(), S, R, f, H, W, self.gs, self.coef.data_ptr(),
                              self.step_dev.data_ptr())

    def _cur(self):
        if self._v is None:   # a program is ~6 GB of packed weights and buffers: never build one as a side effect
            raise hip.RcdmError("DenoiseLoop: no launch plan selected yet — call load() first")
        return self._v

    prog = property(lambda self: self._cur()["prog"])
    _pre = property(lambda self: self._cur()["pre"])
    _post = property(lambda self: self._cur()["post"])

    @property
    def graph(self):
        return self._cur()["graph"]

    @graph.setter
    def graph(self, g):
        self._cur()["graph"] = g

    def _one_step_eager(self):
        for op in self._pre:
            op()
        self.prog.run_body(skip_time=True)
        for op in self._post:
            op()

    def load(self, latents, mask, masked_latents, ctx):
        """Stage the loop inputs in HBM (this is outside the timed hot loop: inputs resident when it starts)."""
        S, R = self.S, self.reps
        assert tuple(latents.shape) == tuple(self.lat.shape), (latents.shape, self.lat.shape)
        assert tuple(mask.shape) == tuple(self.mask.shape), (mask.shape, self.mask.shape)
        assert tuple(masked_latents.shape) == tuple(self.masked.shape)
        self.lat.copy_(latents.to(self.device, torch.float32) * self.init_noise_sigma)
        self.mask.copy_(mask.to(self.device, torch.float32))
        self.masked.copy_(masked_latents.to(self.device, torch.float32))
        # the CFG halves share their UNet input exactly when mask and masked latents repeat (the latents always do)
        share = (self.share_allowed and R == 2 and bool(torch.equal(self.mask[:S], self.mask[S:]))
                 and bool(torch.equal(self.masked[:S], self.masked[S:])))
        if self.split is not None:
            n = S * self.f                          # context rows per CFG half: (R*S*f, L, D), unconditional half first
            ctx = ctx[self.split.half * n:(self.split.half + 1) * n]
        # images whose L context rows are all equal (the reference's unseen frames): their cross-attention is
        # query-independent — the plan variant that skips it, selected per context the way the shared prefix is per story
        runs = None
        if self.rank1_allowed:
            runs = full_rank_runs(ctx.detach().to(self.device))
            if sum(i1 - i0 for i0, i1 in runs) >= ctx.shape[0]:
                runs = None
        self._select(share, runs)
        self.prog.set_context(ctx, force=True)   # ~3 MB + 16 small GEMMs per story: never trust a cache here
        self.step_dev.zero_()
        self._pndm_next = 0
        torch.cuda.current_stream(self.device).synchronize()

    def run(self, callback=None, callback_steps=1, use_graph=True, start=0, steps=None):
        """Run steps [start, start + steps) of the T-step schedule (default: all T) on the program's stream from the
        latents staged by load(); returns the latents (S,4,f,H,W) fp32 (device) after the last step run."""
        p = self.prog
        start = int(start)
        stop = self.T if steps is None else start + int(steps)
        if not (0 <= start < stop <= self.T):
            raise ValueError(f"steps [{start}, {stop}) outside the {self.T}-step schedule")
        if self.pndm and start not in (0, self._pndm_next):
            # the PLMS step is stateful (four prediction slots + the saved first sample in self.hist): a run can only
            # continue where the previous one stopped, or start over
            raise ValueError(f"PNDM: run(start={start}) but the multistep history is valid for step {self._pndm_next} "
                             "(continue there, or start at 0 after load())")
        cur = torch.cuda.current_stream(self.device)
        p.stream.wait_stream(cur)
        with torch.cuda.stream(p.stream):
            if use_graph and self.graph is None:
                # warm every kernel up once outside capture (lazy function loading), then restore the state
                lat0 = self.lat.clone()
                hist0 = self.hist.clone() if self.pndm else None   # (table row 0 overwrites slot 0 and the saved sample)
                self.step_dev.zero_()   # the warm-up step reads table / coefficient row `step`: keep it inside the tables
                self._one_step_eager()
                self.lat.copy_(lat0)
                if hist0 is not None:
                    self.hist.copy_(hist0)
                self.step_dev.zero_()
                p.stream.synchronize()
                self.graph = p.capture(pre=self._pre, post=self._post, skip_time=True)
            self.step_dev.fill_(start)
            for i in range(start, stop):
                if use_graph:
                    self.graph.launch()
                else:
                    self._one_step_eager()
                if callback is not None and i % callback_steps == 0:
                    p.stream.synchronize()
                    callback(i, int(self.timesteps[i]), self.lat)
        self._pndm_next = stop
        cur.wait_stream(p.stream)
        return self.lat


class PriorLoop:
    """The stage-1 sampling loop (reference: `for i, t in enumerate(timesteps)` of Seq_Inpaint_Prior_Pipeline.__call__,
    src/pipelines/prior_pipeline.py:311-344) as T replays of one captured hipGraph: [load t | time embedding | sequence
    assembly | ~600 transformer launches | CFG combine + UnCLIP step | step++].  The scheduler's noise for all T steps is
    drawn up front (from the caller's generator, or supplied), so the captured step is a pure function of device state."""

    def __init__(self, prior, frames, num_text_tokens, guidance_scale, scheduler, num_steps):
        self.prior = prior
        self.gs = float(guidance_scale)
        self.reps = 2 if guidance_scale > 1.0 else 1   # do_classifier_free_guidance, prior_pipeline.py:236-238
        self.n = int(frames)
        self.T = int(num_steps)
        dev = prior.device
        self.device = dev
        if not hasattr(scheduler, "coefficients") or not hasattr(scheduler, "alphas_cumprod"):
            raise NotImplementedError(f"{type(scheduler).__name__}: only rcdms_amd.scheduler.UnCLIPScheduler has a fused step")
        if scheduler.config.prediction_type != "sample":
            raise NotImplementedError("the prior predicts the sample (Kandinsky-2.2 prior scheduler config)")
        scheduler.set_timesteps(self.T, device=None)
        self.timesteps = torch.as_tensor(scheduler.timesteps).to("cpu", torch.int64)
        self.coef = scheduler.coefficients().to(dev)
        self.clip = float(scheduler.config.clip_sample_range) if scheduler.config.clip_sample else 0.0
        self.ts_dev = self.timesteps.to(torch.float32).to(dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.init_noise_sigma = float(getattr(scheduler, "init_noise_sigma", 1.0))
        B = self.reps * self.n
        self.prog = prior._program(B, num_text_tokens)
        E = self.prog.E
        self.lat = torch.zeros(self.n, E, dtype=torch.float32, device=dev)
        self.noise = torch.zeros(self.T, self.n, E, dtype=torch.float32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        p = self.prog
        self._ops = ([lambda: hip.load_timestep(self.ts_dev.data_ptr(), self.step_dev.data_ptr(), p.t_dev.data_ptr(), 1)]
                     + p.step_ops(self.lat, self.n)
                     + [lambda: hip.cfg_unclip_step(p.out16.ptr, p.out16.ld, self.lat.data_ptr(), self.n, self.reps, E,
                                                    self.gs, self.clip, self.coef.data_ptr(), self.noise.data_ptr(),
                                                    self.step_dev.data_ptr()),
                        lambda: hip.advance_step(self.step_dev.data_ptr())])
        self.graph = None
        self._graph_causal = None

    def load(self, latents, proj_embedding, encoder_hidden_states, proj_embedding1, mask_label, attention_mask,
             noise=None, generator=None):
        """Stage the inputs in HBM.  latents (n, E) initial noise; the conditioning tensors already hold the CFG batch
        (B = reps * n rows, unconditional half first, as the reference's _encode_prompt / torch.cat([x] * 2) produce)."""
        assert tuple(latents.shape) == tuple(self.lat.shape), (latents.shape, self.lat.shape)
        self.lat.copy_(latents.to(self.device, torch.float32) * self.init_noise_sigma)
        if noise is None:
            noise = torch.randn(self.noise.shape, dtype=torch.float32, device=self.device, generator=generator)
        self.noise.copy_(noise.to(self.device, torch.float32))
        self.prog.set_context(proj_embedding, encoder_hidden_states, proj_embedding1, mask_label, attention_mask)
        self.step_dev.zero_()
        torch.cuda.current_stream(self.device).synchronize()

    def _one_step_eager(self):
        for op in self._ops:
            op()

    def run(self, use_graph=True):
        """All T steps; returns the final latents (n, E) fp32 on the device (before post_process_latents)."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if use_graph and self.graph is not None and self._graph_causal != self.prog.causal:
                self.graph = None             # the mask mode is baked into the captured launches
            if use_graph and self.graph is None:
                self._graph_causal = self.prog.causal
                lat0 = self.lat.clone()
                self.step_dev.zero_()   # the warm-up step reads table / coefficient row `step`: keep it inside the tables
                self._one_step_eager()        # load every kernel once outside capture, then restore the state
                self.lat.copy_(lat0)
                self.step_dev.zero_()
                self.stream.synchronize()
                g = hip.Graph()
                g.begin()
                try:
                    self._one_step_eager()
                finally:
                    g.end()
                self.stream.synchronize()
                self.step_dev.zero_()
                self.lat.copy_(lat0)
                self.graph = g
            for _ in range(self.T):
                if use_graph:
                    self.graph.launch()
                else:
                    self._one_step_eager()
        cur.wait_stream(self.stream)
        return self.lat
