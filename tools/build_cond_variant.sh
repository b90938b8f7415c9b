#!/bin/bash
# developer tool: a library variant whose whole-stage conditioning kernels (csrc/fastsvc_cond.hip) are compiled with
# extra flags (e.g. -DFASTSVC_COND_TRACE), linked against the cached objects of the product build:
#    tools/build_cond_variant.sh <out.so> <flags...>     then FASTSVC_HIP_LIB=<out.so> python ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$1; shift
cs=$root/svcc23_fastsvc_amd/csrc; b=$root/svcc23_fastsvc_amd/build
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-honor-nans -I $root/include -I $cs"
/opt/rocm/bin/hipcc $common "$@" -x hip -c $cs/fastsvc_cond.hip -o /tmp/cond_var_f32.o &
/opt/rocm/bin/hipcc $common "$@" -DFASTSVC_ACT_BF16=1 -x hip -c $cs/fastsvc_cond.hip -o /tmp/cond_var_bf16.o &
wait
objs=$(ls $b/*.o | grep -v "/cond_" | grep -v "_tl_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/cond_var_f32.o /tmp/cond_var_bf16.o -o $out
echo $out
