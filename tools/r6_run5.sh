cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r6a/wide_tests.log
FASTSVC_WX=2 TL_PER_WAVE=1 FASTSVC_TIMELINE_STORAGE=bfloat16 timeout 600 python tools/timeline.py cfg3 film.2.heads down.3.c2_d2 up.1.d27 > gpurun_out/r6a/timeline_wx.txt 2>&1
tail -n 5 gpurun_out/r6a/wide_tests.log
