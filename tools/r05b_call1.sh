set -x
mkdir -p gpurun_out/r05b
timeout 420 python -m pytest tests/test_project.py tests/test_fused_conv.py tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r05b/pytest_targeted.txt 2>&1; tail -3 gpurun_out/r05b/pytest_targeted.txt
timeout 240 python -m pytest tests/test_step_gpu.py -q -m gpu -k "b32 or b1x1" > gpurun_out/r05b/pytest_step.txt 2>&1; tail -3 gpurun_out/r05b/pytest_step.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/r05b/bench_n1.json 2> gpurun_out/r05b/bench_n1.err; tail -c 400 gpurun_out/r05b/bench_n1.json
timeout 300 bash tools/step_trace.sh r05b/trace > gpurun_out/r05b/trace_stdout.txt 2>&1; tail -5 gpurun_out/r05b/trace_stdout.txt
