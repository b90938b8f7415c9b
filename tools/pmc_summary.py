#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection csv per kernel name: mean of every counter."""
import csv, sys, collections, glob, os
path = sys.argv[1]
files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k.replace("void fastsvc::", "").replace("fastsvc::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(48), " ".join(n[-18:].rjust(18) for n in names), "   n")
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = max(len(v) for v in d.values())
    print(k[:48].ljust(48), " ".join(("%18.4g" % (sum(d[c]) / len(d[c])) if c in d else " " * 18) for c in names), "%4d" % n)
