cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6d/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6d/bench_n1.json 2> gpurun_out/r6d/bench_n1.err
cp bench_detail.json gpurun_out/r6d/bench_detail_n1.json
tail -n 6 gpurun_out/r6d/gpu_tests.log; cut -c1-600 gpurun_out/r6d/bench_n1.json
