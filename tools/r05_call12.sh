set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_parallel.py -q -m gpu -x > gpurun_out/r05/pytest_order_flat.txt 2>&1
tail -5 gpurun_out/r05/pytest_order_flat.txt | cut -c1-200
SCP_ADAMW=torch timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_parallel.py -q -m gpu -x > gpurun_out/r05/pytest_order_torch.txt 2>&1
tail -5 gpurun_out/r05/pytest_order_torch.txt | cut -c1-200
