#!/bin/bash
# per-layer timing under ablation switches (developer tool).  The FASTSVC_DBG switches exist only in the diagnostic
# library (python -m svcc23_fastsvc_amd.build --timeline -> libfastsvc_hip_timeline.so): in the product library the
# never-taken branches themselves cost time.
root=$(cd "$(dirname "$0")/.." && pwd)
export FASTSVC_HIP_LIB=${FASTSVC_HIP_LIB:-$root/svcc23_fastsvc_amd/libfastsvc_hip_timeline.so}
for d in ${DBGS:-0 1 4 5 8}; do
  echo "=== FASTSVC_DBG=$d"
  FASTSVC_DBG=$d python "$root/tools/profile_layers.py" ${1:-cfg2} 2>&1 | grep -E "${LAYERS:-down.0.c123|film.0.chain|down.2.c23|film.3.heads|up.0.d3|up.1.d9|up.2.d9|up.3.up_s|up.3.d3|total}"
done
