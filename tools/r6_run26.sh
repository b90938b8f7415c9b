cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6i/gpu_tests.log
tail -n 6 gpurun_out/r6i/gpu_tests.log
