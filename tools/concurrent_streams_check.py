#!/usr/bin/env python
"""Do two forwards on two HIP streams overlap on this GPU?  (developer tool for the sub-batch experiment)
   python tools/concurrent_streams_check.py [B F]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 600
plan = A.Plan(cfg, compact_workspace=True)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
sets = []
for k in range(2):
    ins = list(S.device_batch(cfg, B, F, 11 + k, dev))
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 1, F * cfg.hop), device=dev)
    sets.append((ins, ws, out))
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

def run(mode, n=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        if mode == "one stream, A then B":
            for ins, ws, out in sets:
                plan.forward(blob, *ins, workspace=ws, out=out)
        else:
            for st, (ins, ws, out) in zip(streams, sets):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    plan.forward(blob, *ins, workspace=ws, out=out)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for mode in ("one stream, A then B", "two streams, A || B"):
    run(mode, 5)
    print(f"B={B} F={F}  {mode}: {run(mode):.3f} ms per pair", flush=True)
