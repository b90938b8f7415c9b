#!/usr/bin/env python
"""Summarise gpurun_out/prof_<tag>/ into profiles/<tag>_* (tracked):
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py`
   <tag>_hbm_traffic.csv    per-kernel-symbol FETCH_SIZE / WRITE_SIZE per launch (separate PMC passes)
   pmc_traffic.json         kernel symbol (bench.py naming) -> HBM bytes per launch (corrected)
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64
bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section), so fetch bytes are doubled."""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "stats_serial", "stats_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "stats_serial", "stats_kernel_stats.csv"),
                os.path.join(dst, f"{tag}_kernel_stats_serial.csv"))


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


fetch = per_kernel(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
rows, js, merged = [], {}, {}
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    if "fastsvc" not in k:
        continue
    f_kib = fetch[k][0] / fetch[k][1]
    w_kib = write[k][0] / write[k][1] if k in write else 0.0
    hbm = (2.0 * f_kib + w_kib) * 1024.0
    rows.append((k, fetch[k][1], f_kib, w_kib, hbm))
    # template arguments: MW, NW, WM, WN, mode, ntaps, epilogue kind, S (polyphase stretch / Winograd
    # dilation, else 1) - bench.py names the kernels by the same eight numbers
    # (a ninth argument, resident weights true / false, does not change the traffic model: merged)
    m = re.search(r"conv_mfma_ws_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>", k)
    if m:
        key = "conv_mfma_ws<%s,%s,%s,%s,%s,%s,%s,%s>" % m.groups()
        n0 = merged.get(key, 0)
        js[key] = (js.get(key, 0.0) * n0 + hbm * fetch[k][1]) / (n0 + fetch[k][1])
        merged[key] = n0 + fetch[k][1]
    # half-precision MFMA kernels: MW, NW, WM, WN, mode, epilogue kind, S (+ resident weights: merged); x3 = split
    # binary16 products (float32 storage), x1 = bf16 products (namespace fastsvc::bf16)
    # (round 3: a tenth argument marks the row-end instances of ragged batches - same traffic model, merged too)
    m = re.search(r"conv_hx_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false)){0,2}>", k)
    if m:
        key = "conv_hx<%s,%s,%s,%s,%s,%s,%s," % m.groups() + ("x1>" if "bf16::" in k else "x3>")
        n0 = merged.get(key, 0)
        js[key] = (js.get(key, 0.0) * n0 + hbm * fetch[k][1]) / (n0 + fetch[k][1])
        merged[key] = n0 + fetch[k][1]
    # the wide-layer kernel (csrc/fastsvc_wx.hip): PRO, EPI template arguments; bench.py names it by shape and epilogue kind
    m = re.search(r"conv_wx_kernel<(\d+), (\d+)>", k)
    if m:
        key = "conv_wx<3,6,4,2,%s,x1>" % m.group(2)
        n0 = merged.get(key, 0)
        js[key] = (js.get(key, 0.0) * n0 + hbm * fetch[k][1]) / (n0 + fetch[k][1])
        merged[key] = n0 + fetch[k][1]
    # whole-stage conditioning launches (csrc/fastsvc_cond.hip)
    m = re.search(r"cond_stage(\d)(_pipe)?_kernel", k)
    if m:
        js["cond_stage%s%s<%s>" % (m.group(1), m.group(2) or "", "x1" if "bf16::" in k else "x3")] = hbm
    m = re.search(r"conv_mfma_kernel<(\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        js["conv_mfma<%s,%s,%s,%s>" % m.groups()] = hbm
    for short in ("in1_conv", "pointwise_out", "spk_proj"):
        if short + "_kernel" in k:
            js[short] = hbm
with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w") as f:
    f.write("kernel,launches_profiled,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch,hbm_bytes_per_launch_corrected\n")
    for r in rows:
        f.write('"%s",%d,%.1f,%.1f,%.0f\n' % r)
# bench.py reads pmc_traffic.json for the cfg2 line's `roofline.traffic`; other workloads get their own file
json.dump(js, open(os.path.join(dst, "pmc_traffic.json" if "cfg" not in tag or "cfg2" in tag else f"{tag}_pmc_traffic.json"), "w"),
          indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:3000])
for r in rows[:8]:
    print("%-70s n=%3d fetch %9.0f KiB write %9.0f KiB -> %7.1f MB/launch" % (r[0][:70], r[1], r[2], r[3], r[4] / 1e6))
