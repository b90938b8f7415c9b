#!/usr/bin/env python
"""Run-to-run determinism of the stage-0 layer pipeline (developer tool): the same forward N times, ss.0 / down_hd.1 of every
run against the first.  FASTSVC_COND_DBG ablation bits apply (results then invalid, but still expected to repeat)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
storage = sys.argv[1] if len(sys.argv) > 1 else "float32"
B, F = int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 52
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
pl = A.Plan(cfg, storage=storage, compact_workspace=True)
T = F * cfg.hop
pl.load_tuned({f"cond.0|{B}|{T}": [1, 1, 1, 1, 5], f"cond.0|{B}|{T}|b": [1, 1, 1, 1, 5], f"cond.1|{B}|{T // 5}|b": [1, 1, 1, 1, 5]})
blob = pl.pack(S.synth_state_dict(cfg, 201)).to(dev)
ins = list(S.device_batch(cfg, B, F, 900 + B, dev))
ws = torch.empty(pl.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
ref = None
nbad = 0
for it in range(int(os.environ.get("RUNS", "12"))):
    ws.fill_(0xFF)
    pl.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize()
    cur = {t: pl.tap(t, B, F, ws).float().clone() for t in ("ss.0", "down_hd.1", "ss.1", "down_hd.2")}
    if ref is None:
        ref = cur
        continue
    for t in cur:
        d = (cur[t] - ref[t]).abs()
        n = int((d > 0).sum())
        if n:
            nbad += 1
            bad = (d > 0).nonzero()
            print(f"  run {it} {t}: {n} elements differ, max {float(d.max()):.3e}, channels {torch.unique(bad[:, 1]).tolist()[:12]}, "
                  f"columns {torch.unique(bad[:, 2]).tolist()[:10]}")
print(f"dbg={os.environ.get('FASTSVC_COND_DBG', '0')} {storage} B={B} F={F}: {nbad} differing (run, tensor) pairs", flush=True)
