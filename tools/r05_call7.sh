set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05/pytest_gpu_full.txt 2>&1
tail -12 gpurun_out/r05/pytest_gpu_full.txt
timeout 600 python tools/phase_timeline.py > gpurun_out/r05/phase_timeline.txt 2>&1
tail -60 gpurun_out/r05/phase_timeline.txt
