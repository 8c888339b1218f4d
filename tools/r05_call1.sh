set -x
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/self-corr-pose_amd
timeout 300 python tools/pk_forms_probe.py > gpurun_out/r05/pk_forms.txt 2>&1
CTL=$GRAFT_REPO_ROOT/self-corr-pose_amd/lib/libscp_hip_slpctl.so
( SCP_HIP_LIB=$CTL ITERS=60 timeout 300 python tools/race_repro.py none mfma_bf16_32 mfma_f32 ) > gpurun_out/r05/ctl_raster.txt 2>&1
( SCP_HIP_LIB=$CTL VICTIM=setup SETUP_PART=fi DIFFS=1 ITERS=300 timeout 300 python tools/race_repro.py mfma_bf16_32 ) > gpurun_out/r05/ctl_setup.txt 2>&1
( ITERS=100 timeout 300 python tools/race_repro.py none mfma_bf16_32 ) > gpurun_out/r05/ship_raster.txt 2>&1
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r05/bench_serial_nopacked.txt 2>&1
( SCP_STREAMS=overlap timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r05/bench_overlap_nopacked.txt 2>&1
tail -3 gpurun_out/r05/*.txt
