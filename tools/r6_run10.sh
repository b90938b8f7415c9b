cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r6c/wide_tests.log
FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6c/wx_v3.txt
FASTSVC_WX=2 TL_PER_WAVE=1 FASTSVC_TIMELINE_STORAGE=bfloat16 timeout 600 python tools/timeline.py cfg3 film.2.heads down.3.c3_d4 up.0.d9 > gpurun_out/r6c/timeline_wx.txt 2>&1
export FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_dbg.so
for d in 4 2 6; do
FASTSVC_DBG=$d FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6c/wx_dbg$d.txt
done
tail -n 12 gpurun_out/r6c/wide_tests.log
