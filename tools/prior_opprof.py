import sys, os, collections, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from rcdms_amd.sampler import PriorLoop
from rcdms_amd.scheduler import UnCLIPScheduler
from src.models.myprior_transformer import MyPriorTransformer
dev = torch.device("cuda", 0)
mk = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"], temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)
with torch.device("meta"):
    m = MyPriorTransformer(num_attention_heads=32, attention_head_dim=64, num_layers=20, embedding_dim=1280, num_embeddings=91, additional_embeddings=6, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_motion_module=True, motion_module_type="Vanilla", motion_module_kwargs=mk)
m = m.to_empty(device=dev).eval(); bench.init_weights_(m)
B,T,E=10,91,1280
g = torch.Generator(device=dev).manual_seed(42)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
loop = PriorLoop(m, 5, T, 4.0, UnCLIPScheduler(), 4)
loop.load(rn(5,E), rn(B,E), rn(B,T,E), rn(B,E), rn(B,E), torch.ones(B,T,device=dev), generator=g)
loop.run(use_graph=False); torch.cuda.synchronize()
plan = loop.prog.plan
agg = collections.defaultdict(lambda: [0, 0.0])
for op, tag in zip(plan.ops, plan.tags):
    op(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): op()
    e1.record(); e1.synchronize()
    r = agg[tag]; r[0] += 1; r[1] += e0.elapsed_time(e1) / 5 * 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.3f} ms over {len(plan.ops)} ops")
for tag, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{us/1e3:8.3f} ms {100*us/tot:5.1f}%  n={n:4d} avg={us/n:8.1f} us  {tag}")
