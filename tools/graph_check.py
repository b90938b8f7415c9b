#!/usr/bin/env python
"""Does replaying the forward as a HIP graph (torch.cuda.CUDAGraph around Plan.forward) pay?"""
import sys, time, torch
sys.path.insert(0, ".")
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
plan = A.Plan(cfg)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
for name in ("cfg1", "cfg2"):
    wl = S.WORKLOADS[name]
    B, F = wl["B"], wl["F"]
    b = S.synth_batch(cfg, B, F, wl["seed"])
    ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 1, F * 160), dtype=torch.float32, device=dev)
    for _ in range(3): plan.forward(blob, *ins, out=out, workspace=ws)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): plan.forward(blob, *ins, out=out, workspace=ws)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 50
    y_eager = out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.forward(blob, *ins, out=out, workspace=ws)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        plan.forward(blob, *ins, out=out, workspace=ws)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 50
    print(f"{name}: eager {eager*1e3:.3f} ms, graph replay {graph*1e3:.3f} ms, max diff {float((out - y_eager).abs().max()):.2e}", flush=True)
