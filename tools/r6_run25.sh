cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
TUNE_ROUNDS=2 TUNE_REPS=2 timeout 2400 python tools/tune_shapes.py cfg3:bf16 cfg2:bf16 > gpurun_out/r6i/tune.log 2>&1
cp svcc23_fastsvc_amd/tuned_mi355x.json gpurun_out/tuned_mi355x.json
timeout 300 python tools/profile_layers.py cfg3 bfloat16 > gpurun_out/r6i/layers_tuned.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r6i/bench.json 2> gpurun_out/r6i/bench.err
tail -n 3 gpurun_out/r6i/tune.log
tail -n 2 gpurun_out/r6i/layers_tuned.txt
head -c 600 gpurun_out/r6i/bench.json
