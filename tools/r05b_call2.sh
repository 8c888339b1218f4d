# tools/r05b_call2.sh -- on the GPU box: the tiled weight-plane kernel (tests, bench, trace) and the N = 2 launch path on one device (gloo)
set -x
mkdir -p gpurun_out/r05c
timeout 300 python -m pytest tests/test_fused_conv.py tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r05c/pytest_conv.txt 2>&1; tail -3 gpurun_out/r05c/pytest_conv.txt
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r05c/bench_n1.json 2> gpurun_out/r05c/bench_n1.err; python -c "import json; d=json.load(open('gpurun_out/r05c/bench_n1.json')); print(d['value'], d['ms_per_step'], d['loss_delta'].get('parity_ok'))"
timeout 300 bash tools/step_trace.sh r05c/trace > gpurun_out/r05c/trace_stdout.txt 2>&1; grep -i "planes\|gradclip" gpurun_out/r05c/trace_kernel_stats_timed_window.csv
SCP_DIST_BACKEND=gloo SCP_SINGLE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05c/bench_n2_gloo_one_device.json 2> gpurun_out/r05c/bench_n2.err; tail -c 300 gpurun_out/r05c/bench_n2_gloo_one_device.json; tail -3 gpurun_out/r05c/bench_n2.err
