#!/usr/bin/env python
"""Pretty-print a bench.py JSON line (developer tool)."""
import json, sys
d = json.load(open(sys.argv[1]))
pk = d["roofline"].pop("per_kernel")
cfg = d.pop("config")
print(json.dumps(d, indent=None)[:1800])
for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("%-28s %8.1f us  x%2d  %6.1f TF  %6.0f GB/s" % (k, v["ms_per_step"] * 1e3, v["launches"], v["TFLOPs"], v["GBs"]))
