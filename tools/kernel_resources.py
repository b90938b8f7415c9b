"""Register / spill table of the kernels in a built object (or every object of the build cache): what hipcc gave each
template instance.  Used to check that a change did not push a variant over its VGPR budget (spills).

    python tools/kernel_resources.py [object.o ...] [--spills]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def table(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("fastsvc::", "").replace("(ConvParams)", "").replace("void ", "")
        out[dem] = dict(vgpr=g("vgpr_count"), vspill=g("vgpr_spill_count"), sgpr=g("sgpr_count"),
                        sspill=g("sgpr_spill_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"))
    return out


if __name__ == "__main__":
    objs = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not objs:
        objs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                             "svcc23_fastsvc_amd", "build", "*.o")))
    for o in objs:
        if "plan" in os.path.basename(o):
            continue
        t = table(o)
        print(f"# {os.path.basename(o)}: {len(t)} kernels")
        for k, v in sorted(t.items()):
            if "--spills" in sys.argv and v["vspill"] == 0:
                continue
            print(f"{k:70s} vgpr {v['vgpr']:4d} spill {v['vspill']:4d} sgpr {v['sgpr']:4d} sspill {v['sspill']:3d} scratch {v['scratch']:5d}")
