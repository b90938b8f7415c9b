#!/usr/bin/env python
"""Stage 0 of the conditioning nets: the layer pipeline (cond_stage0_pipe_kernel, launch-table algorithm 5 under
"cond.0|B|T") against the phase kernel (algorithm 4) - taps ss.0 / down_hd.1, the waveform, and the launch times.
    python tools/cond_pipe_check.py [storage ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

storages = sys.argv[1:] or ["bfloat16", "float32"]
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
SHAPES = [(3, 52, [52, 31, 4]), (1, 300, None), (8, 600, None), (64, 1500, None)]
if os.environ.get("SMALL_ONLY"):
    SHAPES = SHAPES[:2]


def plan_for(storage, B, F, algo):
    pl = A.Plan(cfg, storage=storage, compact_workspace=True)
    T = F * cfg.hop
    pl.load_tuned({f"cond.0|{B}|{T}": [1, 1, 1, 1, algo], f"cond.0|{B}|{T}|b": [1, 1, 1, 1, algo],
                   f"cond.1|{B}|{T // 5}|b": [1, 1, 1, 1, algo]})
    return pl


def prof(plan, blob, ins, ws, lengths=None, n=5):
    tot = {}
    for _ in range(n):
        recs = []
        plan.forward(blob, *ins, workspace=ws, profile=recs, lengths=lengths)
        for r in recs:
            tot[(r["layer"], r["kernel"])] = tot.get((r["layer"], r["kernel"]), 0.0) + r["ms"] / n * 1e3
    return tot


for storage in storages:
    for B, F, lens in SHAPES:
        pipe, phase = plan_for(storage, B, F, 5), plan_for(storage, B, F, 4)
        blob = pipe.pack(S.synth_state_dict(cfg, 201)).to(dev)
        ins = list(S.device_batch(cfg, B, F, 900 + B, dev))
        wp = torch.full((pipe.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)
        wq = torch.full((phase.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)
        for lengths in ([None, None] + ([lens, lens] if lens else [])):
            wp.fill_(0xFF); wq.fill_(0xFF)
            yp = pipe.forward(blob, *ins, workspace=wp, lengths=lengths)
            yq = phase.forward(blob, *ins, workspace=wq, lengths=lengths)
            torch.cuda.synchronize()
            out = []
            for tap in ("ss.0", "down_hd.1", "ss.1", "down_hd.2"):
                a, b = pipe.tap(tap, B, F, wp).float(), phase.tap(tap, B, F, wq).float()
                if lengths is not None:
                    dec = {"ss.0": 1, "down_hd.1": 5, "ss.1": 5, "down_hd.2": 20}[tap]
                    for i, n in enumerate(lengths):          # only the valid columns of every utterance are defined
                        nv = n * cfg.hop // dec
                        for half in range(a.shape[0] // B):
                            a[half * B + i, :, nv:] = 0; b[half * B + i, :, nv:] = 0
                d = (a - b).abs()
                if float(d.max()) > 1e-6 * float(b.abs().max()):
                    bad = (d > 1e-6 * float(b.abs().max())).nonzero()
                    cols = torch.unique(bad[:, 2])
                    print(f"      {tap}: {bad.shape[0]} bad elements, rows {torch.unique(bad[:, 0]).tolist()[:8]}, channels "
                          f"{torch.unique(bad[:, 1]).tolist()[:48]}, {cols.numel()} columns, first {cols[:24].tolist()} last {cols[-8:].tolist()}")
                out.append(f"{tap} max|d| {float(d.max()):.3e} (max {float(b.abs().max()):.3f}, finite {bool(torch.isfinite(a).all())})")
            dy = (yp - yq).abs()
            print(f"{storage} B={B} F={F} lengths={lengths}: " + "  ".join(out) +
                  f"  y max|d| {float(dy.max()):.3e} mean|d| {float(dy.mean()):.2e} finite {bool(torch.isfinite(yp).all())}", flush=True)
        tp, tq = prof(pipe, blob, ins, wp), prof(phase, blob, ins, wq)
        for lay in ("cond.0", "cond.1"):
            kp = [(k, v) for k, v in tp.items() if k[0] == lay][0]
            kq = [(k, v) for k, v in tq.items() if k[0] == lay][0]
            print(f"   {lay}: pipeline {kp[0][1]} {kp[1]:.1f} us   phase {kq[0][1]} {kq[1]:.1f} us   "
                  f"(forward: {sum(tp.values()):.1f} vs {sum(tq.values()):.1f} us)", flush=True)
        del wp, wq, ins
        torch.cuda.empty_cache()
