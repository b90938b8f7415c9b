cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
timeout 600 tools/micro/cu_pull > gpurun_out/r6f/cu_pull.txt 2>&1
cat gpurun_out/r6f/cu_pull.txt
