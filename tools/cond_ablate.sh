#!/bin/bash
# per-phase ablation of the whole-stage conditioning launch (developer tool): time of cond.0 with phases switched off
wl=${1:-cfg2}
for d in 0 1 2 4 8 16 32 63 62 59 47; do
  echo -n "dbg=$d: "
  FASTSVC_COND_DBG=$d python tools/cond_check.py bfloat16 $wl 2>/dev/null | grep "cond.0"
done
