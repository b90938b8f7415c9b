#!/bin/bash
# SQ counters of the whole-stage conditioning launch (developer tool; run through gpurun): tools/cond_pmc.sh [workload]
wl=${1:-cfg2}
out=$GRAFT_REPO_ROOT/gpurun_out/cond_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/cond_check.py ${2:-bfloat16} $wl > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cond_stage" not in k: continue
        k = k.split("(")[0][-28:]
        acc[(k, r["Counter_Name"])]["v"] += float(r["Counter_Value"]); acc[(k, r["Counter_Name"])]["n"] += 1
for c, d in sorted(acc.items()):
    print(f"{str(c):60s} {d['v'] / d['n']:16.0f}  (per launch, {int(d['n'])} launches)")
PY
