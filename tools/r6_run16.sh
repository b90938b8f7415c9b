cd $GRAFT_REPO_ROOT
bash tools/collect_cfg3.sh r6 > gpurun_out/collect_r6.log 2>&1
tail -5 gpurun_out/collect_r6.log
du -sh gpurun_out/prof_r6_cfg3_*
