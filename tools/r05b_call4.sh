# tools/r05b_call4.sh -- on the GPU box: the weight-plane kernel with compile-time kernel sizes and paired stores (tests, smoke, default bench, trace)
set -x
mkdir -p gpurun_out/r05f
timeout 100 python -m pytest tests/test_fused_conv.py tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r05f/pytest_conv.txt 2>&1; tail -2 gpurun_out/r05f/pytest_conv.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r05f/bench_n1.json 2> gpurun_out/r05f/bench_n1.err; python -c "import json; d=json.load(open('gpurun_out/r05f/bench_n1.json')); print(d['value'], d['ms_per_step'])"
timeout 100 bash tools/step_trace.sh r05f/trace > gpurun_out/r05f/trace_stdout.txt 2>&1; grep -i "planes\|gradclip\|multi_tensor" gpurun_out/r05f/trace_kernel_stats_timed_window_all.csv
