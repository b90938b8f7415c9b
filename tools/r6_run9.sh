cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
for d in 0 32 64 96 128 256 384 480; do
FASTSVC_DBG=$d FASTSVC_WX=2 timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6b/wx_exp$d.txt
done
