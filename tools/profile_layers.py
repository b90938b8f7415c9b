#!/usr/bin/env python
"""Per-launch table (hipEvent timed, fastsvc_forward_profile) for one workload on cuda:0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

wl = S.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
storage = sys.argv[2] if len(sys.argv) > 2 else "float32"
plan = A.Plan(cfg, storage=storage, compact_workspace=os.environ.get("FASTSVC_PROFILE_COMPACT", "1") != "0")   # (the layout the module and bench.py run)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
if wl["B"] * wl["F"] > 20000:                       # large workloads: generate on the device
    ins = list(S.device_batch(cfg, wl["B"], wl["F"], wl["seed"], dev))
else:
    b = S.synth_batch(cfg, wl["B"], wl["F"], wl["seed"])
    ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
for _ in range(2):
    plan.forward(blob, *ins, workspace=ws)
if os.environ.get("FASTSVC_AUTOTUNE"):          # time every launch shape first, profile the tuned ones
    plan.forward(blob, *ins, workspace=ws, autotune=True)
N = int(os.environ.get("FASTSVC_PROFILE_N", "5"))
acc = None
for _ in range(N):
    recs = []
    plan.forward(blob, *ins, workspace=ws, profile=recs)
    if acc is None:
        acc = recs
    else:
        for a, r in zip(acc, recs):
            a["ms"] += r["ms"]
tot = sum(a["ms"] for a in acc) / N
print(f"{'layer':22s} {'kernel':16s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>8s} {'%':>5s}")
for a in acc:
    ms = a["ms"] / N
    print(f"{a['layer']:22s} {a['kernel']:16s} {ms*1e3:8.1f} {a['flops']/ms/1e9:8.2f} {a['bytes']/ms/1e6:8.0f} {100*ms/tot:5.1f}")
print(f"total {tot*1e3:.1f} us;  {wl['B']*wl['F']*160/tot/1e3:.1f} Msamples/s (event-timed sum)")
