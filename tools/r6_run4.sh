cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -60 > gpurun_out/r6a/wide_tests.log
for st in 0 3 6 12; do
FASTSVC_WX_STAGGER=$st FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6a/wx_full_st$st.txt
done
export FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_dbg.so
for d in 4 8 12 28; do
FASTSVC_DBG=$d FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6a/wx_dbg$d.txt
done
tail -n 25 gpurun_out/r6a/wide_tests.log
