# tools/r05b_call3.sh -- on the GPU box: FlatAdamW with its per-step scalars as kernel arguments (unit tests, the training-loop tests and a bench line under SCP_ADAMW=flat)
set -x
mkdir -p gpurun_out/r05e
timeout 120 python -m pytest tests/test_project.py -q -m gpu -x > gpurun_out/r05e/pytest_project.txt 2>&1; tail -3 gpurun_out/r05e/pytest_project.txt
SCP_ADAMW=flat timeout 120 python -m pytest tests/test_data.py -q -m gpu -x > gpurun_out/r05e/pytest_data_flat.txt 2>&1; tail -3 gpurun_out/r05e/pytest_data_flat.txt
SCP_ADAMW=flat timeout 150 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r05e/bench_flat.json 2> gpurun_out/r05e/bench_flat.err; python -c "import json; d=json.load(open('gpurun_out/r05e/bench_flat.json')); print(d['value'], d['ms_per_step'], d['gradients'])"
