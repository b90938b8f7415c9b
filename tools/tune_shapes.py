#!/usr/bin/env python
"""Measure the launch-shape table on the GPU in front of us (run through gpurun on an MI355X):

    python tools/tune_shapes.py [cfg1 cfg2 cfg3 ...] -> svcc23_fastsvc_amd/tuned_mi355x.json
                                                      (also copied to gpurun_out/ so it comes back)

For every workload it runs fastsvc_autotune (each pipelined conv times all its candidate tile shapes
and tiles-per-workgroup on the device) REPS times and keeps, per layer, the shape that won most
often; that is done ROUNDS times and the table whose whole forward is fastest is kept (single-launch
timings of close candidates are noisy at the 1-2 % level).  Keys are "<layer>|<B>|<T>", so entries of different workloads do not collide."""
import collections, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
from svcc23_fastsvc_amd.engine import TUNED_TABLE_PATH

REPS = int(os.environ.get("TUNE_REPS", "3"))
names = sys.argv[1:] or ["cfg1", "cfg2"]          # "cfg3:bf16" tunes the bfloat16-storage entries (keys end in "|b"); "32x1500": B x F
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
COMPACT = os.environ.get("TUNE_COMPACT", "1") != "0"        # the layout the module and bench.py run (whole-stage conditioning launches, compact decimated inputs)
ROUNDS = int(os.environ.get("TUNE_ROUNDS", "3"))          # independent tunings per workload; the table that runs the whole forward fastest is kept
# TUNE_INCREMENTAL=1: start from the shipped table and tune only the keys it lacks (a new fused variant's entries)
INCREMENTAL = os.environ.get("TUNE_INCREMENTAL", "0") != "0"
sig = None
table = {}


def time_forward(plan, blob, ins, ws, n=30):
    for _ in range(5):
        plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.forward(blob, *ins, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name in names:
    name, _, st = name.partition(":")
    storage = "bfloat16" if st == "bf16" else "float32"
    if "x" in name and name.split("x")[0].isdigit():          # "32x1500": any batch size x frame count
        wl = {"B": int(name.split("x")[0]), "F": int(name.split("x")[1]), "seed": 99}
    else:
        wl = S.WORKLOADS[name]
    if wl["B"] * wl["F"] > 20000:
        ins = list(S.device_batch(cfg, wl["B"], wl["F"], wl["seed"], dev))
    else:
        b = S.synth_batch(cfg, wl["B"], wl["F"], wl["seed"])
        ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    best = (1e30, None)
    for rnd in range(ROUNDS):
        votes = collections.defaultdict(collections.Counter)
        for rep in range(REPS):
            plan = A.Plan(cfg, load_shipped_table=INCREMENTAL, storage=storage, compact_workspace=COMPACT)
            shipped = dict(plan.tuned_shapes()) if INCREMENTAL else {}
            sig = plan.config_signature()
            blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
            ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
            plan.forward(blob, *ins, workspace=ws)
            plan.forward(blob, *ins, workspace=ws, autotune=True)
            for k, v in plan.tuned_shapes().items():
                if k not in shipped:
                    votes[k][tuple(v)] += 1
        cand = {k: list(c.most_common(1)[0][0]) for k, c in sorted(votes.items())}
        plan = A.Plan(cfg, load_shipped_table=INCREMENTAL, storage=storage, compact_workspace=COMPACT)
        plan.load_tuned(cand)
        if INCREMENTAL:
            print(f"   new entries: {cand}", file=sys.stderr)
        ms = time_forward(plan, blob, ins, ws)
        print(f"{name} ({storage}) tuning {rnd}: {plan.last_autotune_trials or ''} forward {ms:.4f} ms", file=sys.stderr)
        if ms < best[0]:
            best = (ms, cand)
    table.update(best[1])
doc = {"tables": {}}
if os.path.exists(TUNED_TABLE_PATH):
    doc = json.load(open(TUNED_TABLE_PATH))
doc["device"] = torch.cuda.get_device_name(0)
doc["format"] = "tables[config signature][layer|B|T] = [NW, WM, WN, tiles_per_workgroup]"
doc["tables"].setdefault(sig, {}).update(table)
with open(TUNED_TABLE_PATH, "w") as f:                  # one table entry per line
    lines = ["{"] + [" %s: %s," % (json.dumps(k), json.dumps(doc[k])) for k in sorted(doc) if k != "tables"] + [' "tables": {']
    tabs = sorted(doc["tables"].items())
    for ti, (tsig, t) in enumerate(tabs):
        items = sorted(t.items())
        lines.append("  %s: {" % json.dumps(tsig))
        lines += ["   %s: %s%s" % (json.dumps(k), json.dumps(v), "," if i + 1 < len(items) else "") for i, (k, v) in enumerate(items)]
        lines.append("  }%s" % ("," if ti + 1 < len(tabs) else ""))
    f.write("\n".join(lines + [" }", "}"]) + "\n")
os.makedirs("gpurun_out", exist_ok=True)
shutil.copy(TUNED_TABLE_PATH, "gpurun_out/tuned_mi355x.json")
print(f"{len(table)} entries -> {TUNED_TABLE_PATH}")
