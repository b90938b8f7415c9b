#!/bin/bash
# Phase ablation of the stage-0 layer pipeline (developer tool; results of the ablated runs are INVALID, times only).
#   tools/cond_pipe_ablate.sh [storage] [B] [F]
# FASTSVC_COND_DBG bits: 1 no raw-signal loads, 2 no ss stores, 4 no hd, 8 no c1, 16 no matrix layers, 32 no heads
st=${1:-bfloat16}; B=${2:-64}; F=${3:-1500}
for dbg in ${DBG_LIST:-0 1 2 4 8 16 32 6 7 56 63}; do
  FASTSVC_COND_DBG=$dbg FASTSVC_COND_PIPE=2 python - "$st" "$B" "$F" "$dbg" <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
st, B, F, dbg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
pl = A.Plan(cfg, storage=st, compact_workspace=True)
blob = pl.pack(S.synth_state_dict(cfg, 201)).to(dev)
ins = list(S.device_batch(cfg, B, F, 900, dev))
ws = torch.empty(pl.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
t = []
for _ in range(4):
    recs = []
    pl.forward(blob, *ins, workspace=ws, profile=recs)
    t.append([r["ms"] for r in recs if r["layer"] == "cond.0"][0] * 1e3)
k = [r["kernel"] for r in recs if r["layer"] == "cond.0"][0]
print(f"dbg={dbg:3d} {st} B={B} F={F}: cond.0 {k} {min(t[1:]):.1f} us", flush=True)
PY
done
