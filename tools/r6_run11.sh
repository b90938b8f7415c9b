cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r6d/wide_tests.log
FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6d/wx_v2a_forced.txt
TUNE_ROUNDS=2 TUNE_REPS=2 timeout 2400 python tools/tune_shapes.py cfg3:bf16 cfg2:bf16 > gpurun_out/r6d/tune.log 2>&1
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6d/layers_cfg3_bf16_tuned.txt 2>&1
timeout 300 python tools/profile_layers.py cfg2 bfloat16 > gpurun_out/r6d/layers_cfg2_bf16_tuned.txt 2>&1
tail -n 12 gpurun_out/r6d/wide_tests.log; tail -5 gpurun_out/r6d/tune.log
