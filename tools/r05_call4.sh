set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r05/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -c 3000 gpurun_out/r05/bench_default.json
bash tools/step_trace.sh r05/trace > gpurun_out/r05/trace_stdout.txt 2>&1
tail -5 gpurun_out/r05/trace_stdout.txt
