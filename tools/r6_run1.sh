set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r6a/wide_tests.log
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "layer_pipelines or whole_stage or fused_conditioning" 2>&1 | tail -15 > gpurun_out/r6a/cond_tests.log
RUNS=10 timeout 300 python tools/cond_pipe_determinism.py float32 64 1500 > gpurun_out/r6a/determinism.log 2>&1
RUNS=10 timeout 300 python tools/cond_pipe_determinism.py bfloat16 64 1500 >> gpurun_out/r6a/determinism.log 2>&1
timeout 600 python tools/gemm_yardstick.py > gpurun_out/r6a/gemm_yardstick.txt 2>&1
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6a/layers_cfg3_bf16_default.txt 2>&1
FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6a/layers_cfg3_bf16_wx.txt 2>&1
tail -5 gpurun_out/r6a/*.log
