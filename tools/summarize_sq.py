#!/usr/bin/env python
"""profiles/<tag>_sq_counters.csv from gpurun_out/prof_<tag>/pmc_sq (tools/collect_profiles.sh): per kernel
symbol the mean of every SQ counter of the pass, the mean launch duration, and the matrix-pipe
utilisation  SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles)  at the ~2.1 GHz the per-wave
s_memtime stamps measure under this load (tools/timeline.py; DESIGN.md section 6 on DVFS)."""
import collections, csv, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "pmc_sq", "sq_counter_collection.csv")
GHZ, SIMDS = 2.1, 1024
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    if "fastsvc" not in k:
        continue
    k = re.sub(r"\(fastsvc::ConvParams\)$", "", k.replace("void fastsvc::", ""))
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
names = sorted({c for k in acc for c in acc[k]})
rows = []
for k, d in acc.items():
    us = sum(dur[k].values()) / len(dur[k])
    mean = {c: sum(v) / len(v) for c, v in d.items()}
    util = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * us * 1e3 * GHZ)
    rows.append((us * len(dur[k]), k, len(dur[k]), us, util, mean))
rows.sort(reverse=True)
out = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.csv")
with open(out, "w") as f:
    f.write("kernel,launches_profiled,avg_us,mfma_pipe_busy_frac_at_2.1GHz," + ",".join(names) + "\n")
    for _, k, n, us, util, mean in rows:
        f.write('"%s",%d,%.1f,%.3f,' % (k, n, us, util) + ",".join("%.0f" % mean.get(c, 0.0) for c in names) + "\n")
for _, k, n, us, util, mean in rows[:10]:
    print("%-62s n=%3d %7.1f us  MFMA pipe busy %4.1f %%" % (k[:62], n, us, 100 * util))
