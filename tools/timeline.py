#!/usr/bin/env python
"""Per-wave phase timeline of ONE launch of the pipelined convolution kernel (diagnostic).

Needs the stamped build:  python -m svcc23_fastsvc_amd.build --timeline   (-> libfastsvc_hip_timeline.so,
a separate file that only this tool loads; the product library is never stamped).

    python tools/timeline.py cfg2 up.3.d9 [more layers ...]

Each wave's lane 0 writes (tag << 56 | s_memtime) at the phase boundaries of conv_mfma_ws_kernel:
  1 entry   2 first loads / weight stream issued   3 unit 0 committed   4 past the first barrier
  5 producer: next unit staged (global loads waited, transforms done, LDS written)
  6 past the unit barrier      7 consumer: MFMAs of the unit issued     8 consumer: epilogue issued
The summary prints, per role, where the cycles between consecutive stamps go.
"""
import os
import struct
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FASTSVC_HIP_LIB", os.path.join(ROOT, "svcc23_fastsvc_amd", "libfastsvc_hip_timeline.so"))

PER_WAVE = os.environ.get("TL_PER_WAVE") == "1"      # conv_wx: every wave multiplies - roles by wave, not consumer / producer
TAGS = {1: "entry", 2: "issued", 3: "commit0", 4: "bar0", 5: "staged", 6: "bar", 7: "mfma", 8: "epi", 9: "loaded",
        10: "operands", 11: "tile_out"}


def analyse(path, layer):
    raw = open(path, "rb").read()
    cap, wgs, tpw, nchunks, MW, NW, WM, WN = struct.unpack("8q", raw[:64])
    tl = np.frombuffer(raw[64:], dtype=np.uint64).reshape(cap, 8, 64)
    print(f"== {layer}: {wgs} workgroups ({cap} recorded), tpw {tpw}, chunks {nchunks}, shape MW{MW} NW{NW} {WM}x{WN}")
    tag = (tl >> np.uint64(56)).astype(np.int64)
    cyc = (tl & np.uint64((1 << 56) - 1)).astype(np.int64)
    hw = tl[:, :, 62].astype(np.int64)
    xcc = tl[:, :, 63].astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    # workgroup spans and concurrency
    t_in = np.where(tag[:, :, 0] == 1, cyc[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
    last = np.zeros(cap, dtype=np.int64)
    for w in range(cap):
        m = tag[w, :, :62] > 0
        last[w] = cyc[w, :, :62][m].max() if m.any() else 0
    ok = last > 0
    # (s_memtime is per XCD and the eight counters are not synchronised: only differences inside one
    # workgroup are meaningful, so there is no kernel-wide span here)
    span = (last - t_in)[ok]
    print(f"workgroup life: median {int(np.median(span))}, p10 {int(np.percentile(span, 10))}, "
          f"p90 {int(np.percentile(span, 90))} cyc")
    cus = set(zip(xcc[ok, 0], se[ok, 0], sh[ok, 0], cu[ok, 0]))
    print(f"distinct (xcc, se, sh, cu) seen: {len(cus)}; simd of wave 0..7 in wg 0: {simd[0].tolist()}")
    # residency: workgroups of one CU whose lives overlap (same XCD -> same s_memtime base)
    by_cu = defaultdict(list)
    for w in np.flatnonzero(ok):
        by_cu[(int(xcc[w, 0]), int(se[w, 0]), int(sh[w, 0]), int(cu[w, 0]))].append((int(t_in[w]), int(last[w])))
    peak = []
    for spans in by_cu.values():
        ev = sorted([(a, 1) for a, _ in spans] + [(b, -1) for _, b in spans])
        cur = best = 0
        for _, d in ev:
            cur += d
            best = max(best, cur)
        peak.append(best)
    print(f"workgroups resident together on one CU: max {max(peak)}, median {int(np.median(peak))}")
    # phase durations per role
    seg = defaultdict(list)
    for w in np.flatnonzero(ok)[:2048]:
        for wave in range(8):
            tg, cy = tag[w, wave, :62], cyc[w, wave, :62]
            n = int((tg > 0).sum())
            role = f"wave{wave}" if PER_WAVE else ("cons" if wave < 4 else "prod")
            for i in range(1, n):
                seg[(role, TAGS.get(int(tg[i - 1]), "?"), TAGS.get(int(tg[i]), "?"))].append(int(cy[i] - cy[i - 1]))
    tot = defaultdict(int)
    for (role, a, b), v in seg.items():
        tot[role] += sum(v)
    print(f"{'role':5s} {'from':8s} {'to':8s} {'count':>8s} {'median':>8s} {'mean':>8s} {'p90':>8s} {'share':>6s}")
    for (role, a, b), v in sorted(seg.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
        v = np.array(v)
        print(f"{role:5s} {a:8s} {b:8s} {len(v):8d} {int(np.median(v)):8d} {int(v.mean()):8d} "
              f"{int(np.percentile(v, 90)):8d} {100 * v.sum() / tot[role]:5.1f}%")
    # one example workgroup, relative cycles
    w = int(np.flatnonzero(ok)[len(np.flatnonzero(ok)) // 2])
    print(f"example workgroup {w} (cycles since its entry):")
    for wave in ((0, 4, 5, 7) if PER_WAVE else (0, 4)):
        tg, cy = tag[w, wave, :62], cyc[w, wave, :62]
        n = int((tg > 0).sum())
        print(f"  wave {wave}: " + " ".join(f"{TAGS.get(int(tg[i]), '?')}@{int(cy[i] - t_in[w])}" for i in range(min(n, 40))))


def main():
    import torch
    import svcc23_fastsvc_amd as A
    from svcc23_fastsvc_amd import synth as S
    wl = S.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    layers = sys.argv[2:] or ["up.3.d9"]
    cfg = S.FULL_CONFIG
    dev = torch.device("cuda:0")
    storage = os.environ.get("FASTSVC_TIMELINE_STORAGE", "float32")
    plan = A.Plan(cfg, storage=storage, compact_workspace=True)
    if os.environ.get("TL_TABLE"):                  # extra launch-table entries as JSON (e.g. another tpw for the layer looked at)
        import json
        plan.load_tuned(json.loads(os.environ["TL_TABLE"]))
    blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
    if wl["B"] * wl["F"] > 20000:
        ins = list(S.device_batch(cfg, wl["B"], wl["F"], wl["seed"], dev))
    else:
        b = S.synth_batch(cfg, wl["B"], wl["F"], wl["seed"])
        ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
    os.makedirs("gpurun_out", exist_ok=True)
    for _ in range(3):
        plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize()
    for layer in layers:
        out = f"gpurun_out/timeline_{layer}.bin"
        os.environ["FASTSVC_TIMELINE_LAYER"] = layer
        os.environ["FASTSVC_TIMELINE_OUT"] = out
        plan.forward(blob, *ins, workspace=ws)
        torch.cuda.synchronize()
        if not os.path.exists(out):
            print(f"{layer}: no such pipelined launch (or not a --timeline build)")
            continue
        analyse(out, layer)
        os.remove(out)


if __name__ == "__main__":
    main()
