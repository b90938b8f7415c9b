#!/usr/bin/env python
"""Whole-stage conditioning launch (csrc/fastsvc_cond.hip) against the separate launches (developer tool):
    python tools/cond_check.py [storage] [workload ...]
taps ss.0 / down_hd.1 and the waveform of a compact-workspace plan (fused) against a default-layout plan (separate
launches, every tap inspectable), then the per-launch tables of both."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

storage = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
names = sys.argv[2:] or ["cfg1", "cfg2"]
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
fused, sep = A.Plan(cfg, storage=storage, compact_workspace=True), A.Plan(cfg, storage=storage)
blob = fused.pack(S.synth_state_dict(cfg, 201)).to(dev)


def prof(plan, ins, ws, n=5):
    acc = None
    for _ in range(n):
        recs = []
        plan.forward(blob, *ins, workspace=ws, profile=recs)
        if acc is None:
            acc = recs
        else:
            for a, r in zip(acc, recs):
                a["ms"] += r["ms"]
    return {a["layer"]: a["ms"] / n * 1e3 for a in acc}


for name in names:
    wl = S.WORKLOADS[name]
    B, F = wl["B"], wl["F"]
    ins = list(S.device_batch(cfg, B, F, wl["seed"], dev))
    wsf = torch.full((fused.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)
    wss = torch.empty(sep.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    yf = fused.forward(blob, *ins, workspace=wsf)
    ys = sep.forward(blob, *ins, workspace=wss)
    torch.cuda.synchronize()
    ssf, sss = fused.tap("ss.0", B, F, wsf).float(), sep.tap("ss.0", B, F, wss).float()
    hdf = fused.tap("down_hd.1", B, F, wsf).float()
    hds = sep.tap("down_h.0", B, F, wss).float()[..., ::5]
    s1f, s1s = fused.tap("ss.1", B, F, wsf).float(), sep.tap("ss.1", B, F, wss).float()
    h2f, h2s = fused.tap("down_hd.2", B, F, wsf).float(), sep.tap("down_h.1", B, F, wss).float()[..., ::4]
    print(f"{name} {storage}: stage 1: ss.1 max|d| {float((s1f - s1s).abs().max()):.3e} (max {float(s1s.abs().max()):.3f}, mean|d| "
          f"{float((s1f - s1s).abs().mean()):.2e})  hd.2 max|d| {float((h2f - h2s).abs().max()):.3e} (max {float(h2s.abs().max()):.3f})")
    print(f"{name} {storage}: ss.0 max|d| {float((ssf - sss).abs().max()):.3e} (max|ss| {float(sss.abs().max()):.3f}, "
          f"mean|d| {float((ssf - sss).abs().mean()):.2e})  hd max|d| {float((hdf - hds).abs().max()):.3e} "
          f"(max {float(hds.abs().max()):.3f})  y max|d| {float((yf - ys).abs().max()):.3e} mean|d| "
          f"{float((yf - ys).abs().mean()):.2e} (max|y| {float(ys.abs().max()):.3f})  finite {bool(torch.isfinite(yf).all())}",
          flush=True)
    pf, ps = prof(fused, ins, wsf), prof(sep, ins, wss)
    only = [k for k in pf if k not in ps] + [k for k in ps if k not in pf]
    for k in list(pf) + [k for k in ps if k not in pf]:
        if k in only or abs(pf.get(k, 0) - ps.get(k, 0)) > 0.15 * max(pf.get(k, 0), ps.get(k, 0)):
            print(f"   {k:22s} fused {pf.get(k, float('nan')):9.1f} us   separate {ps.get(k, float('nan')):9.1f} us")
    print(f"   total: fused {sum(pf.values()):.1f} us, separate {sum(ps.values()):.1f} us", flush=True)
    del wsf, wss
