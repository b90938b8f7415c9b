for st in bfloat16 float32; do
for k in 0 938 469 313 235 118; do
FASTSVC_COND_TPW=$k python - $st $k <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
st, k = sys.argv[1], sys.argv[2]
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
pl = A.Plan(cfg, storage=st, compact_workspace=True)
blob = pl.pack(S.synth_state_dict(cfg, 201)).to(dev)
ins = list(S.device_batch(cfg, 64, 1500, 900, dev))
ws = torch.empty(pl.workspace_bytes(64, 1500), dtype=torch.uint8, device=dev)
t = {}
for _ in range(5):
    recs = []
    pl.forward(blob, *ins, workspace=ws, profile=recs)
    for r in recs:
        if r["layer"].startswith("cond."): t.setdefault(r["layer"], []).append(r["ms"] * 1e3)
print(st, "COND_TPW", k, {a: round(min(v[1:]), 1) for a, v in t.items()}, flush=True)
PY
done; done
for k in 0 375 188 94 47; do
FASTSVC_COND1_TPW=$k python - bfloat16 $k <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
st, k = sys.argv[1], sys.argv[2]
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
pl = A.Plan(cfg, storage=st, compact_workspace=True)
blob = pl.pack(S.synth_state_dict(cfg, 201)).to(dev)
ins = list(S.device_batch(cfg, 64, 1500, 900, dev))
ws = torch.empty(pl.workspace_bytes(64, 1500), dtype=torch.uint8, device=dev)
t = {}
for _ in range(5):
    recs = []
    pl.forward(blob, *ins, workspace=ws, profile=recs)
    for r in recs:
        if r["layer"].startswith("cond."): t.setdefault(r["layer"], []).append(r["ms"] * 1e3)
print(st, "COND1_TPW", k, {a: round(min(v[1:]), 1) for a, v in t.items()}, flush=True)
PY
done
