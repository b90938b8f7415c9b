#!/usr/bin/env python
"""How much do frame counts that are not a multiple of 4 cost? (the T/80- and T/160-rate tensors then
have rows that are not float4-aligned)"""
import sys, time, torch
sys.path.insert(0, ".")
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
plan = A.Plan(cfg)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
for B in (1, 8):
    for F in (600, 601, 602, 603):
        b = S.synth_batch(cfg, B, F, 77)
        ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
        ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        for _ in range(3): plan.forward(blob, *ins, workspace=ws)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): plan.forward(blob, *ins, workspace=ws)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print(f"B={B} F={F}: {dt*1e3:.3f} ms  {B*F*160/dt/1e6:.1f} Msamples/s", flush=True)
