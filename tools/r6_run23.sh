cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
DBG_LIST="0 2 4 6 8 16 32 48 24 40 56 22 38 54 62 63" timeout 900 bash tools/cond_pipe_ablate.sh bfloat16 64 1500 > gpurun_out/r6h/cond0_ablate.txt 2>&1
FASTSVC_TIMELINE_STORAGE=bfloat16 timeout 600 python tools/timeline.py cfg3 up.3.d3x up.3.d9 > gpurun_out/r6h/timeline_up3.txt 2>&1
grep "dbg=" gpurun_out/r6h/cond0_ablate.txt
