#!/usr/bin/env python
"""Time chosen layers of cfg2 under FORCED launch shapes (developer tool): loads table entries for the
layers below with every (NW, tiles-per-workgroup) in the sweep and prints the per-layer hipEvent times.
Used for A/B runs of kernel variants the tuner would not pick (e.g. FASTSVC_NO_... switches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
wl = S.WORKLOADS["cfg2"]; cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
b = S.synth_batch(cfg, wl["B"], wl["F"], wl["seed"])
ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
for nw in (1, 2):
    for tpw in (4, 8, 12):
        plan = A.Plan(cfg)
        blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
        ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
        T = wl["F"] * 160
        plan.load_tuned({f"down.0.c2_d2|8|{T}": [nw, 1, 4, tpw, 0], f"film.0.conv|8|{T}": [nw, 1, 4, tpw, 0], f"down.0.c3_d4|8|{T}": [nw, 1, 4, tpw, 0]})
        for _ in range(2): plan.forward(blob, *ins, workspace=ws)
        acc = None
        for _ in range(5):
            recs = []
            plan.forward(blob, *ins, workspace=ws, profile=recs)
            if acc is None: acc = recs
            else:
                for a, r in zip(acc, recs): a["ms"] += r["ms"]
        out = {a["layer"]: (a["kernel"], a["ms"] / 5 * 1e3) for a in acc if a["layer"] in ("down.0.c2_d2", "film.0.conv", "down.0.c3_d4")}
        print(nw, tpw, " ".join(f"{k}={v[1]:.1f}" for k, v in out.items()), list(out.values())[0][0])
