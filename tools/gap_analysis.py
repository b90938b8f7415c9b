#!/usr/bin/env python
"""Idle time between the kernels of one forward, from a rocprofv3 kernel trace (diagnostic).

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py --steps 5 --warmup 3 ...
    python tools/gap_analysis.py OUT [launches_per_step]

Takes the LAST complete step (launches_per_step kernels ending with pointwise_out), prints every kernel with its
start relative to the step, duration, queue, and the gap to the previous kernel end on the critical chain, then the
union-busy time of the device against the step's wall time."""
import csv, glob, sys

d = sys.argv[1]
path = [p for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(path)))
name_k = "Kernel_Name"; s_k = "Start_Timestamp"; e_k = "End_Timestamp"
rows.sort(key=lambda r: int(r[s_k]))
ends = [i for i, r in enumerate(rows) if "pointwise_out" in r[name_k]]
if len(ends) < 2:
    sys.exit("fewer than two forwards in the trace")
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = int(step[0][s_k])
print(f"{len(step)} kernels in the last step; columns: start_us dur_us gap_us queue kernel")
busy_end = t0
idle = 0
for r in step:
    s, e = int(r[s_k]), int(r[e_k])
    gap = s - busy_end
    if gap > 0:
        idle += gap
    nm = r[name_k]
    nm = nm[nm.find("conv_"):][:60] if "conv_" in nm else nm[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {nm}")
    busy_end = max(busy_end, e)
wall = busy_end - t0
print(f"wall {wall / 1e3:.1f} us; device idle (no kernel running) {idle / 1e3:.1f} us = {100.0 * idle / wall:.1f} %")
