cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r6e/wide_tests.log
timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6e/wx_v4.txt
TL_PER_WAVE=1 FASTSVC_TIMELINE_STORAGE=bfloat16 timeout 600 python tools/timeline.py cfg3 film.2.heads down.3.c3_d4 up.0.d9 > gpurun_out/r6e/timeline_wx.txt 2>&1
tail -n 6 gpurun_out/r6e/wide_tests.log
