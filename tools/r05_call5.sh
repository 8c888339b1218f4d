set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
for i in 1 2 3; do timeout 300 python -m pytest tests/test_render_golden.py -q -m gpu 2>&1 | tail -3; done > gpurun_out/r05/render_alone.txt 2>&1
SCP_STREAMS=serial timeout 300 python -m pytest tests/test_render_golden.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r05/render_alone.txt
cat gpurun_out/r05/render_alone.txt
timeout 2400 python -m pytest tests/test_softras_gpu.py tests/test_softras_ref_gpu.py tests/test_split_accuracy_gpu.py tests/test_step_gpu.py tests/test_trainer_host.py tests/test_vit_gpu.py tests/test_fused_conv.py tests/test_pretrained_golden.py -q -m gpu > gpurun_out/r05/pytest_gpu_rest.txt 2>&1
tail -25 gpurun_out/r05/pytest_gpu_rest.txt
