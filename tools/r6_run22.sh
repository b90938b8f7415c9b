cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 300 python tools/d3x_check.py > gpurun_out/r6g/d3x_check.txt 2>&1
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_wide_gpu.py -x -q -k "bf16 or bfloat16 or wide or folded" 2>&1 | tail -30 > gpurun_out/r6g/tests2.log
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6g/layers_w8c.txt 2>&1
tail -n 12 gpurun_out/r6g/d3x_check.txt
tail -n 6 gpurun_out/r6g/tests2.log
grep "d3x\|total" gpurun_out/r6g/layers_w8c.txt
