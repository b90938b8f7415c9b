cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r6b/wide_tests.log
FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6b/wx_v2.txt
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6b/layers_default.txt 2>&1
FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_stnt.so timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6b/layers_stnt.txt 2>&1
FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_ldstnt.so timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6b/layers_ldstnt.txt 2>&1
timeout 300 python tools/profile_layers.py cfg3 float32 > gpurun_out/r6b/layers_default_f32.txt 2>&1
FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_stnt.so timeout 300 python tools/profile_layers.py cfg3 float32 > gpurun_out/r6b/layers_stnt_f32.txt 2>&1
FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_ldstnt.so timeout 300 python tools/profile_layers.py cfg3 float32 > gpurun_out/r6b/layers_ldstnt_f32.txt 2>&1
tail -n 12 gpurun_out/r6b/wide_tests.log
