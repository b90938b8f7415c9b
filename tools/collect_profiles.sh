#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun):
#   tools/collect_profiles.sh r1      -> gpurun_out/prof_r1/{stats,pmc_fetch,pmc_write}
# Counters are collected in their own passes with --kernel-trace only (MI355X guide).
tag=${1:-r1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/stats.log 2>&1
# same command with the helper streams disabled: every kernel runs alone, which is how bench.py's
# per-kernel roofline pass (fastsvc_forward_profile) times them
FASTSVC_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/stats_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg2 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg2 > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o sq -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg2 > $out/pmc_sq.log 2>&1
find $out -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
