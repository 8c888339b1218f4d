set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 900 python -m pytest tests/test_step_gpu.py -q -m gpu -s -k "b32 or b8" > gpurun_out/r05/step_b32_verbose.txt 2>&1
grep -E 'rel L2|rel |passed|failed|max abs dev' gpurun_out/r05/step_b32_verbose.txt | cut -c1-160 | tail -70
# the N > 1 launch path on one device (gloo, both ranks on cuda:0): smoke of bench.py --gpus 2 with the fused clip (prescale 1 / world)
SCP_DIST_BACKEND=gloo SCP_SINGLE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 4 --warmup 2 --no-isolated > gpurun_out/r05/bench_2ranks_one_device.txt 2>&1
tail -c 1500 gpurun_out/r05/bench_2ranks_one_device.txt
timeout 600 python -m pytest tests/test_render_golden.py tests/test_graphed_gpu.py tests/test_trainer_host.py -q -m gpu 2>&1 | tail -3
