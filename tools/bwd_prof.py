#!/usr/bin/env python
"""Where the PyTorch side of a training step goes (developer tool): the differentiable restatement of the dataflow
(autograd.py) forward and forward + backward at the recipe batch, with a torch.profiler table."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S, autograd as AG
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
g = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels), upsampling_scales=list(cfg.upsampling_scales), out_channels=cfg.out_channels, spk_emb_size=cfg.spk_emb_size, use_spk_emb=cfg.use_spk_emb)
g.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 7).items()}, strict=True)
g = g.train().to(dev)
ins = list(S.device_batch(cfg, 32, 100, 11, dev))
names = [n for n, _ in g.named_parameters()]; params = [p for _, p in g.named_parameters()]
def fw():
    w = AG.folded_weights(dict(zip(names, params)))
    return AG._forward_torch(w, g.upsampling_scales, ins[0], ins[1], ins[2], ins[3])
for _ in range(3):
    y = fw(); y.mean().backward()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): y = fw()
torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 10
t = time.perf_counter()
for _ in range(10):
    y = fw(); y.mean().backward()
torch.cuda.synchronize(); tfb = (time.perf_counter() - t) / 10
print(f"torch restatement: forward {tf*1e3:.1f} ms, forward+backward {tfb*1e3:.1f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    y = fw(); y.mean().backward(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
