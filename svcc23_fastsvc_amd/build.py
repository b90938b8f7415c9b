"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

    python -m svcc23_fastsvc_amd.build

Output: svcc23_fastsvc_amd/libfastsvc_hip.so (git-ignored, travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libfastsvc_hip.so")
# the stamped diagnostic build (--timeline) is a separate file: loaded only when FASTSVC_HIP_LIB names it
TIMELINE_LIB_PATH = os.path.join(PKG_DIR, "libfastsvc_hip_timeline.so")
SOURCES = ["fastsvc_kernels.hip", "fastsvc_hx.hip", "fastsvc_wx.hip", "fastsvc_cond.hip", "fastsvc_plan.cpp", "fastsvc_signal.hip", "fastsvc_loudness.hip", "fastsvc_stftloss.hip", "fastsvc_convgrad.hip", "fastsvc_filmnorm.hip", "fastsvc_gconv.hip"]
HEADERS = [os.path.join(CSRC, "fastsvc_kernels.h"), os.path.join(ROOT, "include", "fastsvc_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 kernels)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


# (source, extra flags, object name): fastsvc_kernels.hip is compiled twice - float32 and bfloat16
# activation storage (namespace fastsvc / fastsvc::bf16) - the objects are built in parallel
UNITS = [
    ("fastsvc_kernels.hip", [], "kernels_f32.o"),
    ("fastsvc_kernels.hip", ["-DFASTSVC_ACT_BF16=1"], "kernels_bf16.o"),
    ("fastsvc_hx.hip", [], "hx_f32.o"),
    ("fastsvc_hx.hip", ["-DFASTSVC_ACT_BF16=1"], "hx_bf16.o"),
    ("fastsvc_wx.hip", [], "wx_f32.o"),
    ("fastsvc_wx.hip", ["-DFASTSVC_ACT_BF16=1"], "wx_bf16.o"),
    # (-fno-honor-nans: LeakyReLU as max(v, 0.2 v) without the canonicalising v_max v, v, v in front of it)
    ("fastsvc_cond.hip", ["-fno-honor-nans"], "cond_f32.o"),
    ("fastsvc_cond.hip", ["-fno-honor-nans", "-DFASTSVC_ACT_BF16=1"], "cond_bf16.o"),
    ("fastsvc_plan.cpp", [], "plan.o"),
    ("fastsvc_signal.hip", [], "signal.o"),
    ("fastsvc_loudness.hip", [], "loudness.o"),
    ("fastsvc_stftloss.hip", [], "stftloss.o"),
    ("fastsvc_convgrad.hip", [], "convgrad.o"),
    ("fastsvc_filmnorm.hip", [], "filmnorm.o"),
    ("fastsvc_stage.hip", [], "stage.o"),
    ("fastsvc_gconv.hip", [], "gconv.o"),
]


def build(force: bool = False, verbose: bool = False, timeline: bool = False) -> str:
    """timeline=True: diagnostic build with per-wave cycle stamps in the pipelined kernel
    (-DFASTSVC_TIMELINE, tools/timeline.py) written to libfastsvc_hip_timeline.so, which only
    tools/timeline.py loads (FASTSVC_HIP_LIB); the product library is left untouched."""
    if not force and not timeline and not os.environ.get("FASTSVC_BUILD_OUT") and not needs_build():
        return LIB_PATH
    import hashlib
    hipcc = _hipcc()
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    # developer A/B builds: extra -D switches and another output file (tools only; FASTSVC_HIP_LIB loads it)
    extra_flags = os.environ.get("FASTSVC_BUILD_FLAGS", "").split()
    common += extra_flags
    if timeline:
        common.append("-DFASTSVC_TIMELINE=1")
        common.append("-DFASTSVC_DEBUG_SWITCHES=1")      # FASTSVC_DBG ablation switches exist in this build only
    # object cache (git-ignored build/ directory): a unit is recompiled only when its source, any header
    # under csrc/ or include/, or its flags changed - one kernel file takes minutes, the host file seconds
    cache = os.path.join(PKG_DIR, "build")
    os.makedirs(cache, exist_ok=True)
    hdr_hash = hashlib.sha1()
    for h in sorted(HEADERS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]):
        hdr_hash.update(open(h, "rb").read())
    procs, objs = [], []
    for src, extra, obj in UNITS:
        key = hashlib.sha1(open(os.path.join(CSRC, src), "rb").read() + hdr_hash.digest() +
                           " ".join(common + extra).encode()).hexdigest()[:16]
        # (developer A/B builds keep their own objects: they must not evict the product build's from the cache)
        stem = os.path.splitext(obj)[0] + ("_tl" if timeline else "") + ("_ab" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:6] if extra_flags else "")
        path = os.path.join(cache, f"{stem}_{key}.o")
        objs.append(path)
        if os.path.exists(path) and not force:
            continue
        for stale in os.listdir(cache):
            if stale.startswith(stem + "_") and stale[len(stem) + 1:len(stem) + 17] != key and \
                    len(stale) == len(stem) + 19 and stale.endswith(".o"):
                os.remove(os.path.join(cache, stale))
        cmd = [hipcc, *common, *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", path + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, path, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, path, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out)
        os.replace(path + ".tmp", path)
    out_path = os.environ.get("FASTSVC_BUILD_OUT") or (TIMELINE_LIB_PATH if timeline else LIB_PATH)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", out_path + ".tmp"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(out_path + ".tmp", out_path)
    return out_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timeline="--timeline" in sys.argv))
