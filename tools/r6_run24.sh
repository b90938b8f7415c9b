cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
DBG_LIST="${DBGS:-0 32 64}" timeout 600 bash tools/cond_pipe_ablate.sh bfloat16 64 1500 > gpurun_out/r6h/cond0_ablate_hcol.txt 2>&1
grep "dbg=" gpurun_out/r6h/cond0_ablate_hcol.txt
