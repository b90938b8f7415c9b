cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r6a/wide_tests.log
FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6a/layers_cfg3_bf16_wx.txt 2>&1
tail -n 8 gpurun_out/r6a/wide_tests.log
