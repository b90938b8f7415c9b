cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6g/layers_w8.txt 2>&1
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_wide_gpu.py -x -q -k "bf16 or bfloat16 or wide" 2>&1 | tail -30 > gpurun_out/r6g/tests.log
tail -n 8 gpurun_out/r6g/tests.log
tail -n 3 gpurun_out/r6g/layers_w8.txt
