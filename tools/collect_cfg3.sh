#!/bin/bash
# rocprofv3 evidence for BASELINE config 3 (64 x 10 s) in both activation storages (run through gpurun):
#   tools/collect_cfg3.sh r2   -> gpurun_out/prof_r2_cfg3_{float32,bfloat16}/{stats,pmc_fetch,pmc_write,pmc_sq}
# Counters in their own passes with --kernel-trace only (MI355X guide).
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
export FASTSVC_PROFILE_N=2
for st in float32 bfloat16; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_cfg3_$st
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --storage $st --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg3 $st > $out/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg3 $st > $out/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o sq -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg3 $st > $out/pmc_sq.log 2>&1
  # raw traces are large: keep the summaries only
  find $out -name "*kernel_trace.csv" -delete
  find $out -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
done
