#!/bin/bash
# per-layer timing under ablation switches (developer tool)
for d in ${DBGS:-0 1 2 4 6 7 16}; do
  echo "=== FASTSVC_DBG=$d"
  FASTSVC_DBG=$d python tools/profile_layers.py ${1:-cfg2} 2>&1 | grep -E "${LAYERS:-down.0.c2|film.0.heads|down.2.c2|film.3.heads|up.0.d3|up.1.d9|up.2.d9|up.3.up_s|up.3.d3|total}"
done
