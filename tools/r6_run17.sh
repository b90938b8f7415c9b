cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
for st in 0 1 2 4 8; do
FASTSVC_WX_STAGGER=$st timeout 300 python tools/profile_layers.py cfg3 bfloat16 2>&1 | grep "conv_wx\|total" > gpurun_out/r6e/wx_st$st.txt
done
