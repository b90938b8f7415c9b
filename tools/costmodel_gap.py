"""Per layer: the launch shape the static cost model picks for an off-table (B, F) against the one on-device
autotuning picks, with the serial time of each (fastsvc_forward_profile).   python tools/costmodel_gap.py 30 540 [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

B, F = int(sys.argv[1]), int(sys.argv[2])
storage = "bfloat16" if len(sys.argv) > 3 and sys.argv[3].startswith("b") else "float32"
ragged = "ragged" in sys.argv            # profile with per-utterance lengths (F - 2 frames each)
kw = {"lengths": [F - 2] * B} if ragged else {}
dev = torch.device("cuda:0")
cfg = S.FULL_CONFIG
ins = list(S.device_batch(cfg, B, F, 5, dev))
res = {}
for mode in ("model", "tuned"):
    plan = A.Plan(cfg, compact_workspace=True, storage=storage)
    blob = plan.pack(S.synth_state_dict(cfg, 1)).to(dev)
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    if mode == "tuned":
        plan.forward(blob, *ins, workspace=ws, autotune=True)
    for _ in range(2):
        plan.forward(blob, *ins, workspace=ws, profile=[], **kw)            # (first launches load the code objects)
    agg = {}
    for _ in range(5):
        recs = []
        plan.forward(blob, *ins, workspace=ws, profile=recs, **kw)
        for r in recs:
            a = agg.setdefault(r["layer"], [r["kernel"], 0.0])
            a[1] += r["ms"] / 5
    res[mode] = agg
    import time
    for _ in range(3): plan.forward(blob, *ins, workspace=ws, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): plan.forward(blob, *ins, workspace=ws, **kw)
    torch.cuda.synchronize(); res[mode + "_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    shapes = plan.tuned_shapes()
    res[mode + "_shapes"] = shapes
tot_m = sum(v[1] for v in res["model"].values()); tot_t = sum(v[1] for v in res["tuned"].values())
print(f"B {B} F {F} {storage}{' ragged' if ragged else ''}: serial sum cost model {tot_m:.3f} ms, autotuned {tot_t:.3f} ms; "
      f"forward (multi-stream) {res['model_ms']:.3f} / {res['tuned_ms']:.3f} ms")
rows = []
for layer, (k, ms) in res["model"].items():
    kt, mst = res["tuned"].get(layer, ("-", 0.0))
    rows.append((ms - mst, layer, k, ms, kt, mst))
for d, layer, k, ms, kt, mst in sorted(rows, reverse=True)[:24]:
    tl = [v for kk, v in res["tuned_shapes"].items() if kk.split("|")[0] == layer and int(kk.split("|")[1]) == B]
    print(f"{layer:22s} model {k:30s} {ms*1e3:8.1f} us | tuned {kt:30s} {mst*1e3:8.1f} us  {tl[0] if tl else ''}")
