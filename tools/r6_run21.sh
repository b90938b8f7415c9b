cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r6g/wide_tests.log
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6g/layers_w8b.txt 2>&1
tail -n 6 gpurun_out/r6g/wide_tests.log
grep "conv_wx\|total" gpurun_out/r6g/layers_w8b.txt
