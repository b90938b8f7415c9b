#!/usr/bin/env python
"""VERDICT r2 task 7: do utterance sub-batches whose inter-kernel tensors fit the 256 MB Infinity Cache run faster
per utterance than the whole batch?  Times one forward of B x 10 s for several B (cost-model launch shapes unless the
table has the batch size) and prints ms per utterance.   python tools/subbatch_mall_check.py [float32|bfloat16]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
storage = sys.argv[1] if len(sys.argv) > 1 else "float32"
plan = A.Plan(cfg, storage=storage, compact_workspace=True)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
F = 1500
for B in (2, 4, 8, 16, 32, 64):
    ins = list(S.device_batch(cfg, B, F, 7, dev))
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    for tune in (False, True):
        if tune:
            plan.forward(blob, *ins, workspace=ws, autotune=True)
        for _ in range(3): plan.forward(blob, *ins, workspace=ws)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(max(2, 64 // B)): plan.forward(blob, *ins, workspace=ws)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / max(2, 64 // B))
        print(f"{storage} B={B:2d} x 10 s {'autotuned ' if tune else 'table/model'}: {best:8.3f} ms per forward, {best / B:7.4f} ms per utterance, x64 = {best / B * 64:7.2f} ms", flush=True)
    del ws, ins
    torch.cuda.empty_cache()
