cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 600 python -m pytest tests/test_wide_gpu.py -q 2>&1 | grep -n "^E  \|passed\|failed\|Error" | cut -c1-250 | head -30 > gpurun_out/r6d/wide.log
cat gpurun_out/r6d/wide.log
