cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 600 python -m pytest tests/test_parity_gpu.py -q -k "near_constant" 2>&1 | grep -n "assert\|Error\|passed\|failed" | head -20 > gpurun_out/r6d/nc.log
FASTSVC_COND_PIPE=0 timeout 600 python -m pytest tests/test_parity_gpu.py -q -k "near_constant" 2>&1 | grep -n "assert\|Error\|passed\|failed" | head -20 >> gpurun_out/r6d/nc.log
cat gpurun_out/r6d/nc.log
