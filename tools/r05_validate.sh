#!/bin/bash
# tools/r05_validate.sh -- on the GPU box: the round's validation run (full GPU suite, smoke, default bench line, kernel trace, PMC traffic),
# then the stall reproducer with the opt-in optimizer and the stream probe (tools/hang_probe.py)
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out/r05; cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=20 > gpurun_out/r05/pytest_gpu_final.txt 2>&1
tail -8 gpurun_out/r05/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
tail -c 600 gpurun_out/r05/bench_n1.json
bash tools/step_trace.sh r05/final > gpurun_out/r05/trace_final_stdout.txt 2>&1
bash tools/traffic_run.sh > gpurun_out/r05/traffic_final_stdout.txt 2>&1
if [ "$1" = "probe" ]; then
    PYTHONPATH=tools SCP_ADAMW=flat SCP_PROBE_AFTER=150 timeout 1000 python -m pytest -p hang_probe tests/test_conv_gpu.py \
        tests/test_coresidency_gpu.py tests/test_corr.py tests/test_data.py -q -m gpu -x > gpurun_out/r05/hang_probe_flat.txt 2>&1
    grep -n "passed\|failed\|hang_probe\|idle=\|event busy\|events:\|GPU use" gpurun_out/r05/hang_probe_flat.txt | head -60
fi
