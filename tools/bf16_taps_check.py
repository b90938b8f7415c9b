#!/usr/bin/env python
"""Developer check: per-tap deviation of the bfloat16-storage forward from the float32-storage one
(same weights / inputs), to localise a wrong kernel variant.  python tools/bf16_taps_check.py [B F]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

B, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 48)
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
sd = S.synth_state_dict(cfg, 81)
b = S.synth_batch(cfg, B, F, 82)
ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
p32, p16 = A.Plan(cfg), A.Plan(cfg, storage="bfloat16")
p32.keep_last_block_output(B, F); p16.keep_last_block_output(B, F)      # the up.3.out tap
blob = p32.pack(sd).to(dev)
w32 = torch.zeros(p32.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
w16 = torch.zeros(p16.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
recs = []
y32 = p32.forward(blob, *ins, workspace=w32)
y16 = p16.forward(blob, *ins, workspace=w16, profile=recs)
kern = {r["layer"]: r["kernel"] for r in recs}
names = []
n = cfg.n_stages
for k in range(n):
    names += [f"down_c1.{k}", f"down_c2.{k}", f"down_h.{k}", f"film_u.{k}", f"ss.{k}"]
    if k: names.insert(-5, f"down_r.{k}")
for i in range(n):
    names += [f"up.{i}.{t}" for t in ("a", "xr", "u1", "xmid", "u2", "u3", "out")]
for name in names:
    a, c = p32.tap(name, B, F, w32).float(), p16.tap(name, B, F, w16).float()
    err = (a - c).abs()
    print(f"{name:14s} max {float(err.max()):9.3e} mean {float(err.mean()):9.3e}  (|ref| max {float(a.abs().max()):8.3f})")
print("y", float((y32 - y16).abs().max()), float((y32 - y16).abs().mean()))
for k, v in kern.items():
    print(f"  {k:20s} {v}")
