#!/bin/bash
# tools/r06_first_call.sh -- what round 5 could not run any more (its GPU budget ended): on the GPU box
#   1. the second revision of the batched weight-plane kernel: bit equality on the encoder's layer list and its time
#   2. the full GPU suite under SCP_ADAMW=flat (FlatAdamW with kernel-argument scalars): the suite stalled twice with the FIRST version of
#      that optimizer; two clean runs of this one are the condition for making it the default
#   3. bench lines of both optimizers
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out/r06a; cd $R
timeout 200 python tools/planes_bench.py > gpurun_out/r06a/planes_bench.txt 2>&1; tail -3 gpurun_out/r06a/planes_bench.txt
SCP_ADAMW=flat timeout ${SUITE_TIMEOUT:-900} python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r06a/pytest_gpu_adamw_flat.txt 2>&1
tail -15 gpurun_out/r06a/pytest_gpu_adamw_flat.txt
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
SCP_ADAMW=flat timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_adamw_flat.json 2> gpurun_out/r06a/bench_adamw_flat.err
python -c "
import json
for n in ('default', 'adamw_flat'):
    d = json.load(open('gpurun_out/r06a/bench_%s.json' % n)); print(n, d['value'], d['ms_per_step'])"
