#!/usr/bin/env python
"""Per-kernel time of ONE steady-state train step from a rocprofv3 kernel trace (csv): the launches between the last two
`stft_loss_finalize_kernel` launches (one per step), grouped by kernel name.
    python tools/step_breakdown.py <kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "stft_loss_finalize_kernel" in r["Kernel_Name"]]
# (the bench's kernel micro-benchmarks launch the loss on its own afterwards: a STEP is a pair of marks with thousands of
# launches in between - the last such pair)
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 1000]
a, b = pairs[-2] if len(pairs) > 1 else pairs[-1]      # (the last one runs into the micro-benchmarks)
acc, cnt = defaultdict(float), defaultdict(int)
for r in rows[a:b]:
    n = r["Kernel_Name"]
    acc[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[n] += 1
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
tot = sum(acc.values())
print(f"step wall {wall:.0f} us, kernel sum {tot:.0f} us, {b - a} launches")
for n, t in sorted(acc.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{t:9.1f} us {cnt[n]:5d}x  {n[:150]}")
