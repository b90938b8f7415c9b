#!/usr/bin/env python
"""Forward time of the shipped-table plan on synthetic workloads (developer tool):
    python tools/fwd_time.py [cfg1 cfg2 cfg3 cfg3:bf16 ...]      best of 5 x 50 forwards after 10 warm-up ones"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
plans = {}
for name in sys.argv[1:] or ["cfg1", "cfg2"]:
    name, _, st = name.partition(":")
    storage = "bfloat16" if st == "bf16" else "float32"
    if storage not in plans:
        plan = A.Plan(cfg, storage=storage, compact_workspace=True)
        plans[storage] = (plan, plan.pack(S.synth_state_dict(cfg, 201)).to(dev))
    plan, blob = plans[storage]
    wl = S.WORKLOADS[name]
    ins = list(S.device_batch(cfg, wl["B"], wl["F"], wl["seed"], dev))
    ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
    for _ in range(10): plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): plan.forward(blob, *ins, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    print(f"{name} ({storage}): {best:.4f} ms", flush=True)
