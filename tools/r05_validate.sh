#!/bin/bash
# tools/r05_validate.sh -- on the GPU box: the round's validation run (full GPU suite, smoke, default bench line, kernel trace, PMC traffic)
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out/r05; cd $R
timeout 2700 python -m pytest tests -q -m gpu --durations=20 > gpurun_out/r05/pytest_gpu_final.txt 2>&1
tail -8 gpurun_out/r05/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
tail -c 600 gpurun_out/r05/bench_n1.json
bash tools/step_trace.sh r05/final > gpurun_out/r05/trace_final_stdout.txt 2>&1
bash tools/traffic_run.sh > gpurun_out/r05/traffic_final_stdout.txt 2>&1
