#!/usr/bin/env python
"""Cost of one generator training step through the module mirror (developer tool; SURVEY 8 f2 slice):
forward on the HIP path (re-folding and re-packing the weights the optimizer just changed), backward by PyTorch-ROCm
autograd over the restatement, RAdam-like update.  Batch 32 x 100 frames = the recipe's batch (fastsvc.yaml)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
g = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                       upsampling_scales=list(cfg.upsampling_scales), out_channels=cfg.out_channels,
                       spk_emb_size=cfg.spk_emb_size, use_spk_emb=cfg.use_spk_emb)
g.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 7).items()}, strict=True)
g = g.train().to(dev)
opt = torch.optim.Adam(g.parameters(), lr=1e-5)
B, F = 32, 100
ins = list(S.device_batch(cfg, B, F, 11, dev))


def step():
    y = g(*ins)
    loss = (y * y).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
parts = {}
for name in ("pack", "forward", "backward+update"):
    parts[name] = 0.0
n = 10
t0 = time.perf_counter()
for _ in range(n):
    ta = time.perf_counter()
    g.packed_weights(dev); torch.cuda.synchronize()
    tb = time.perf_counter()
    y = g(*ins); torch.cuda.synchronize()
    tc = time.perf_counter()
    loss = (y * y).mean(); opt.zero_grad(); loss.backward(); opt.step(); torch.cuda.synchronize()
    td = time.perf_counter()
    parts["pack"] += tb - ta; parts["forward"] += tc - tb; parts["backward+update"] += td - tc
total = (time.perf_counter() - t0) / n
print(f"train step (B={B}, F={F}): {total * 1e3:.1f} ms  = " +
      ", ".join(f"{k} {v / n * 1e3:.1f} ms" for k, v in parts.items()))
