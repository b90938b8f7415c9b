cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
echo new > gpurun_out/r6d/nc_sweep.log
timeout 900 python tools/near_constant_sweep.py 2>&1 | grep -v Warn | tail -3 >> gpurun_out/r6d/nc_sweep.log
echo old >> gpurun_out/r6d/nc_sweep.log
FASTSVC_HIP_LIB=$GRAFT_REPO_ROOT/svcc23_fastsvc_amd/libfastsvc_hip_oldcond.so timeout 900 python tools/near_constant_sweep.py 2>&1 | grep -v Warn | tail -3 >> gpurun_out/r6d/nc_sweep.log
cat gpurun_out/r6d/nc_sweep.log
