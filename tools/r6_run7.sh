cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
FASTSVC_WX=2 TL_PER_WAVE=1 FASTSVC_TIMELINE_STORAGE=bfloat16 timeout 600 python tools/timeline.py cfg3 film.2.heads down.3.c3_d4 up.0.d9 > gpurun_out/r6b/timeline_wx.txt 2>&1
