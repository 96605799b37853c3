"""Ordered op list of one denoising step of the bench workload (launch plan tags).  usage: python tools/dump_plan.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from rcdms_amd import synth
from rcdms_amd.sampler import DenoiseLoop
from rcdms_amd.scheduler import DDIMScheduler
model = bench.build_model(torch.device("cuda", 0))
sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
loop = DenoiseLoop(model, 1, 5, 64, 64, 85, 2.0, sched, 4)
story = synth.synthetic_story(stories=1, latent_hw=(64, 64), ctx_len=85, seed=42)
loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
for i, t in enumerate(loop.prog.plan.tags):
    print(i, t)
