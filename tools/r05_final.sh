#!/bin/bash
# tools/r05_final.sh -- on the GPU box: the round's closing run on the final tree: smoke, default bench line (+ CPU baseline), kernel trace
# with the full kernel table, free-running phase timeline, then the whole GPU suite
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out/r05d; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/r05d/bench_n1.json 2> gpurun_out/r05d/bench_n1.err
python -c "import json; d=json.load(open('gpurun_out/r05d/bench_n1.json')); print(d['value'], d['ms_per_step'], d['loss_delta'].get('parity_ok'), d['roofline']['frac'])"
timeout 200 bash tools/step_trace.sh r05d/trace > gpurun_out/r05d/trace_stdout.txt 2>&1
grep -i "planes\|gradclip\|multi_tensor" gpurun_out/r05d/trace_kernel_stats_timed_window_all.csv
timeout 120 python tools/phase_timeline.py --lookahead > gpurun_out/r05d/phase_timeline.txt 2>&1; tail -3 gpurun_out/r05d/phase_timeline.txt
timeout ${SUITE_TIMEOUT:-840} python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05d/pytest_gpu_final.txt 2>&1
tail -25 gpurun_out/r05d/pytest_gpu_final.txt
